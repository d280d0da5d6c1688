// Probe: is `v_sub_f32 ... div:2` (VOP3 output modifier) bit-identical to `(a - b) * 0.5f` under the
// strict flavour's float mode (f32 denormals: inputs honoured, results flushed), and does the
// output modifier need the MODE.IEEE bit cleared?  Prints mismatch counts per variant.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Xclang -fdenormal-fp-math-f32=preserve-sign,ieee \
//         -o omod_probe omod_probe.hip && ./omod_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

template <bool CLEAR_IEEE>
__global__ void probe(const float *a, const float *b, float *plain, float *omod, int n)
{
    if (CLEAR_IEEE) __builtin_amdgcn_s_setreg(1 | (9 << 6) | (0 << 11), 0); // hwreg(HW_REG_MODE, 9, 1) = IEEE
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    const float d = x - y;
    plain[i] = d * 0.5f;
    float r;
    asm("v_sub_f32_e64 %0, %1, %2 div:2" : "=v"(r) : "v"(x), "v"(y));
    omod[i] = r;
}

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

int main()
{
    std::vector<float> a, b;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> uni(0.f, 1.f);
    for (int i = 0; i < 1 << 20; ++i) { a.push_back(uni(rng)); b.push_back(uni(rng)); }
    // tiny differences, sub-normal operands and results, flush boundary, huge values, specials
    for (int i = 0; i < 1 << 18; ++i) {
        const uint32_t e = rng() % 40;          // exponents 0..39 (sub-normal .. 2^-88)
        const uint32_t m1 = rng() & 0x7fffff, m2 = rng() & 0x7fffff;
        a.push_back(from_bits((e << 23) | m1));
        b.push_back(from_bits(((e + (rng() % 3) - 1u) << 23 & 0x7f800000u) | m2));
    }
    for (int i = 0; i < 1 << 16; ++i) {
        a.push_back(from_bits((rng() & 0x807fffffu) | (254u << 23)));
        b.push_back(from_bits((rng() & 0x807fffffu) | ((253u + (rng() & 1)) << 23)));
    }
    const float sp[] = {0.f, -0.f, INFINITY, -INFINITY, NAN, 1.f, -1.f, from_bits(1), from_bits(0x00800000), from_bits(0x00ffffff),
                        from_bits(0x01000000), from_bits(0x807fffff), 3.4028235e38f};
    for (float x : sp) for (float y : sp) { a.push_back(x); b.push_back(y); }
    const int n = (int)a.size();
    float *da, *db, *dp, *dm;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dp, n * 4); hipMalloc(&dm, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> p(n), m(n);
    for (int variant = 0; variant < 2; ++variant) {
        hipMemset(dm, 0xff, n * 4);
        if (variant == 0) probe<false><<<(n + 255) / 256, 256>>>(da, db, dp, dm, n);
        else probe<true><<<(n + 255) / 256, 256>>>(da, db, dp, dm, n);
        hipDeviceSynchronize();
        hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
        long bad = 0, bad_nonnan = 0, unscaled = 0, zero_sign = 0;
        for (int i = 0; i < n; ++i) {
            if (bits(p[i]) != bits(m[i])) {
                ++bad;
                if ((bits(p[i]) | 0x80000000u) == 0x80000000u && (bits(m[i]) | 0x80000000u) == 0x80000000u) {
                    ++zero_sign; // a flushed result: -0 from the multiply, +0 from the output modifier
                    continue;
                }
                if (!(std::isnan(p[i]) && std::isnan(m[i]))) {
                    if (bad_nonnan < 8)
                        std::printf("  a=%08x b=%08x plain=%08x omod=%08x\n", bits(a[i]), bits(b[i]), bits(p[i]), bits(m[i]));
                    ++bad_nonnan;
                }
                if (m[i] == a[i] - b[i] && m[i] != 0.f) ++unscaled;
            }
        }
        std::printf("IEEE bit %s: %d cases, %ld mismatches (%ld differ only in the sign of a zero, %ld other non-NaN, "
                    "%ld where omod was ignored)\n",
                    variant ? "cleared" : "as compiled", n, bad, zero_sign, bad_nonnan, unscaled);
    }
    return 0;
}
