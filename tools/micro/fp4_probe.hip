// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands +-4 and block scales 2^4 x 2^4 as an exact Hamming engine
// (a product is -4096 where the bits agree, +4096 where they differ; 64 bits per instruction).  Diagnostics, not part of the library:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fp4_probe tools/micro/fp4_probe.hip && /tmp/fp4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
// one wave: A = 32 "trains" x 64 bits, B = 32 "queries" x 64 bits; out[q][t] = dot
__global__ void probe(const uint64_t* trains, const uint64_t* queries, float* out, int sa, int sb) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    // lane (r, h): 32 fp4 values = bits 32 h .. 32 h + 31 of row r, as nibbles: train +4 (0x6) set / -4 (0xE) clear; query -4 set / +4 clear
    auto expand = [&](uint64_t bits, bool train) {
        v8i v = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 4; w++) {
            unsigned x = 0;
            for (int n = 0; n < 8; n++) {
                const int bit = (int)((bits >> (32 * h + 8 * w + n)) & 1);
                const unsigned nib = (train ? bit : !bit) ? 0x6u : 0xEu;
                x |= nib << (4 * n);
            }
            v[w] = (int)x;
        }
        return v;
    };
    const v8i a = expand(trains[r], true), b = expand(queries[r], false);
    v16f c;
    for (int g = 0; g < 16; g++) c[g] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, sa, 0, sb);
    for (int g = 0; g < 16; g++) {
        const int t = (g & 3) + 8 * (g >> 2) + 4 * h;      // A row
        out[r * 32 + t] = c[g];                             // B column = lane & 31
    }
}
int main() {
    uint64_t ht[32], hq[32];
    srand(1);
    for (int i = 0; i < 32; i++) { ht[i] = ((uint64_t)rand() << 33) ^ ((uint64_t)rand() << 11) ^ rand(); hq[i] = ((uint64_t)rand() << 35) ^ ((uint64_t)rand() << 9) ^ rand(); }
    uint64_t *dt, *dq; float* dout;
    hipMalloc(&dt, sizeof(ht)); hipMalloc(&dq, sizeof(hq)); hipMalloc(&dout, 4 * 1024);
    hipMemcpy(dt, ht, sizeof(ht), hipMemcpyHostToDevice); hipMemcpy(dq, hq, sizeof(hq), hipMemcpyHostToDevice);
    const int sa = 127 + 4, sb = 127 + 4;        // E8M0: 2^4 each (byte 0 of the scale operand)
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dt, dq, dout, sa, sb);
    float ho[1024];
    hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int q = 0; q < 32; q++)
        for (int t = 0; t < 32; t++) {
            const int ham = __builtin_popcountll(ht[t] ^ hq[q]);
            const float want = 4096.f * (2 * ham - 64);
            if (ho[q * 32 + t] != want) { if (bad < 5) printf("q %d t %d: got %.1f want %.1f (hamming %d)\n", q, t, ho[q * 32 + t], want, ham); bad++; }
        }
    printf("%s: %d of 1024 wrong\n", bad ? "MISMATCH" : "exact", bad);
    return bad != 0;
}
