// TEST INFRASTRUCTURE: the slice of thrust::device_vector the reference's simple_knn.cu uses (simple_knn.cu:194-216), on the host
#pragma once
#include <vector>
namespace thrust {
template <class T> struct device_ptr_stub { T* p; T* get() const { return p; } };
template <class T>
struct device_vector {
    std::vector<T> v;
    device_vector() {}
    explicit device_vector(size_t n) : v(n) {}
    device_ptr_stub<T> data() { return device_ptr_stub<T>{ v.data() }; }
    typename std::vector<T>::iterator begin() { return v.begin(); }
    typename std::vector<T>::iterator end() { return v.end(); }
    size_t size() const { return v.size(); }
    void resize(size_t n) { v.resize(n); }
};
}
