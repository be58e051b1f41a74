/*
 * glmlike.h -- TEST INFRASTRUCTURE (part of oracle/).  Not shipped, not linked by the product.
 *
 * Minimal column-major vec/mat types that reproduce, operation by operation, the
 * evaluation order of the GLM 0.9.9.9 operators the reference kernels use
 * (reference: submodules/diff-gaussian-rasterization/third_party/glm/glm/detail/
 *   type_mat3x3.inl:468-519  (mat3*vec3, vec3*mat3, mat3*mat3),
 *   type_mat4x4.inl:630-648  (mat4*mat4),
 *   func_matrix.inl:30-37    (outerProduct), :119-137 (transpose),
 *   func_geometric.inl:8-14, 48-55 (length, dot)).
 * Everything is written as plain left-to-right fp32 expressions; the oracle is compiled
 * with -ffp-contract=off so no multiply-add is ever fused.  oracle/check_glm.cpp compares
 * these helpers bit-for-bit against the vendored GLM headers where /root/reference exists.
 */
#ifndef GOF_ORACLE_GLMLIKE_H
#define GOF_ORACLE_GLMLIKE_H
#include <cmath>

namespace gl {

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

// m[c][r]: column c, row r (GLM storage)
struct M3 {
    float m[3][3];
    float* operator[](int c) { return m[c]; }
    const float* operator[](int c) const { return m[c]; }
};
struct M4 {
    float m[4][4];
    float* operator[](int c) { return m[c]; }
    const float* operator[](int c) const { return m[c]; }
};

// glm::mat3(x0,y0,z0, x1,y1,z1, x2,y2,z2): arguments fill columns
static inline M3 mat3(float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2)
{
    M3 r; r[0][0] = x0; r[0][1] = y0; r[0][2] = z0; r[1][0] = x1; r[1][1] = y1; r[1][2] = z1; r[2][0] = x2; r[2][1] = y2; r[2][2] = z2; return r;
}
static inline M4 mat4(float x0, float y0, float z0, float w0, float x1, float y1, float z1, float w1,
                      float x2, float y2, float z2, float w2, float x3, float y3, float z3, float w3)
{
    M4 r;
    r[0][0] = x0; r[0][1] = y0; r[0][2] = z0; r[0][3] = w0;
    r[1][0] = x1; r[1][1] = y1; r[1][2] = z1; r[1][3] = w1;
    r[2][0] = x2; r[2][1] = y2; r[2][2] = z2; r[2][3] = w2;
    r[3][0] = x3; r[3][1] = y3; r[3][2] = z3; r[3][3] = w3;
    return r;
}

// type_mat3x3.inl:486-519
static inline M3 mul(const M3& a, const M3& b)
{
    M3 r;
    for (int c = 0; c < 3; c++)
        for (int row = 0; row < 3; row++)
            r[c][row] = a[0][row] * b[c][0] + a[1][row] * b[c][1] + a[2][row] * b[c][2];
    return r;
}
// type_mat3x3.inl:468-474  (m * v)
static inline V3 mul(const M3& m, const V3& v)
{
    return V3{ m[0][0] * v.x + m[1][0] * v.y + m[2][0] * v.z,
               m[0][1] * v.x + m[1][1] * v.y + m[2][1] * v.z,
               m[0][2] * v.x + m[1][2] * v.y + m[2][2] * v.z };
}
// type_mat3x3.inl:477-483  (v * m)
static inline V3 mul(const V3& v, const M3& m)
{
    return V3{ m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z,
               m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
               m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z };
}
// type_mat4x4.inl:630-648: Result[c] = ((A0*Bc[0] + A1*Bc[1]) + A2*Bc[2]) + A3*Bc[3], per component
static inline M4 mul(const M4& a, const M4& b)
{
    M4 r;
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++)
            r[c][row] = a[0][row] * b[c][0] + a[1][row] * b[c][1] + a[2][row] * b[c][2] + a[3][row] * b[c][3];
    return r;
}
static inline M3 transpose(const M3& m)
{
    M3 r;
    for (int c = 0; c < 3; c++) for (int row = 0; row < 3; row++) r[c][row] = m[row][c];
    return r;
}
static inline M4 transpose(const M4& m)
{
    M4 r;
    for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) r[c][row] = m[row][c];
    return r;
}
static inline M3 neg(const M3& m)
{
    M3 r;
    for (int c = 0; c < 3; c++) for (int row = 0; row < 3; row++) r[c][row] = -m[c][row];
    return r;
}
static inline M3 add(const M3& a, const M3& b)
{
    M3 r;
    for (int c = 0; c < 3; c++) for (int row = 0; row < 3; row++) r[c][row] = a[c][row] + b[c][row];
    return r;
}
// func_matrix.inl:30-37: m[i] = c * r[i]
static inline M3 outerProduct(const V3& c, const V3& r)
{
    M3 m;
    const float rr[3] = { r.x, r.y, r.z };
    for (int i = 0; i < 3; i++) { m[i][0] = c.x * rr[i]; m[i][1] = c.y * rr[i]; m[i][2] = c.z * rr[i]; }
    return m;
}
static inline float dot(const V3& a, const V3& b)
{
    const float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return tx + ty + tz;
}
static inline float length(const V3& v) { return sqrtf(dot(v, v)); }
static inline V3 operator+(const V3& a, const V3& b) { return V3{ a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline V3 operator-(const V3& a, const V3& b) { return V3{ a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline V3 operator-(const V3& a) { return V3{ -a.x, -a.y, -a.z }; }
static inline V3 operator*(float s, const V3& a) { return V3{ s * a.x, s * a.y, s * a.z }; }
static inline V3 operator*(const V3& a, float s) { return V3{ a.x * s, a.y * s, a.z * s }; }
static inline V3 operator/(const V3& a, float s) { return V3{ a.x / s, a.y / s, a.z / s }; }

} // namespace gl
#endif
