// check_glm.cpp -- TEST INFRASTRUCTURE.  Bit-for-bit comparison of oracle/glmlike.h against the
// GLM 0.9.9.9 headers vendored by the reference (third_party/glm), on random inputs.
// Built and run only where /root/reference exists (tests/test_oracle_pins.py); prints "OK n".
#include <cstdio>
#include <cstring>
#include <random>
#include <glm/glm.hpp>
#include "glmlike.h"

static bool same(float a, float b) { return memcmp(&a, &b, 4) == 0; }
static int bad = 0;
#define CHECK(a, b) do { if (!same((a), (b))) { bad++; if (bad < 10) printf("mismatch line %d: %.9g vs %.9g\n", __LINE__, (double)(a), (double)(b)); } } while (0)

int main()
{
    std::mt19937 rng(123);
    std::normal_distribution<float> nd(0.f, 3.f);
    int n = 0;
    for (int it = 0; it < 20000; it++) {
        float a[16], b[16], v[3], w[3];
        for (auto& x : a) x = nd(rng);
        for (auto& x : b) x = nd(rng);
        for (auto& x : v) x = nd(rng);
        for (auto& x : w) x = nd(rng);
        glm::mat3 A(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]), B(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8]);
        gl::M3 A2 = gl::mat3(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8]), B2 = gl::mat3(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8]);
        glm::vec3 V(v[0], v[1], v[2]), Wv(w[0], w[1], w[2]);
        gl::V3 V2{ v[0], v[1], v[2] }, W2{ w[0], w[1], w[2] };
        glm::mat3 AB = A * B; gl::M3 AB2 = gl::mul(A2, B2);
        glm::mat3 ABt = glm::transpose(A) * glm::transpose(B) * A; gl::M3 ABt2 = gl::mul(gl::mul(gl::transpose(A2), gl::transpose(B2)), A2);
        glm::mat3 S = A + glm::outerProduct(V, Wv); gl::M3 S2 = gl::add(A2, gl::outerProduct(V2, W2));
        for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { CHECK(AB[c][r], AB2[c][r]); CHECK(ABt[c][r], ABt2[c][r]); CHECK(S[c][r], S2[c][r]); }
        glm::vec3 AV = A * V, VA = V * A, nAV = -A * V, nVA = -V * A;
        gl::V3 AV2 = gl::mul(A2, V2), VA2 = gl::mul(V2, A2), nAV2 = gl::mul(gl::neg(A2), V2), nVA2 = gl::mul(-V2, A2);
        CHECK(AV.x, AV2.x); CHECK(AV.y, AV2.y); CHECK(AV.z, AV2.z);
        CHECK(VA.x, VA2.x); CHECK(VA.y, VA2.y); CHECK(VA.z, VA2.z);
        CHECK(nAV.x, nAV2.x); CHECK(nAV.y, nAV2.y); CHECK(nAV.z, nAV2.z);
        CHECK(nVA.x, nVA2.x); CHECK(nVA.y, nVA2.y); CHECK(nVA.z, nVA2.z);
        CHECK(glm::length(V), gl::length(V2));
        CHECK(glm::dot(V, Wv), gl::dot(V2, W2));
        glm::vec3 D = V / glm::length(V), E = 0.3f * v[0] * Wv, F = Wv * 0.7f * v[1];
        gl::V3 D2 = V2 / gl::length(V2), E2 = 0.3f * v[0] * W2, F2 = W2 * 0.7f * v[1];
        CHECK(D.x, D2.x); CHECK(D.y, D2.y); CHECK(D.z, D2.z); CHECK(E.x, E2.x); CHECK(E.z, E2.z); CHECK(F.y, F2.y);
        glm::mat4 M(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
        glm::mat4 N(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], b[12], b[13], b[14], b[15]);
        gl::M4 M2 = gl::mat4(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
        gl::M4 N2 = gl::mat4(b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], b[12], b[13], b[14], b[15]);
        glm::mat4 MN = M * N, MtN = glm::transpose(M) * N; gl::M4 MN2 = gl::mul(M2, N2), MtN2 = gl::mul(gl::transpose(M2), N2);
        for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) { CHECK(MN[c][r], MN2[c][r]); CHECK(MtN[c][r], MtN2[c][r]); }
        n++;
    }
    if (bad) { printf("FAILED %d mismatches\n", bad); return 1; }
    printf("OK %d\n", n);
    return 0;
}
