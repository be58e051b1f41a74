#pragma once
#include <numeric>
namespace thrust { template <class It> inline void sequence(It a, It b) { std::iota(a, b, 0); } }
