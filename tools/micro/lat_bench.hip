// Latency of the building blocks of a Jacobi rotation on ONE wave (diagnostics, not part of the library):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/lat_bench tools/micro/lat_bench.hip && /tmp/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cfloat>
#include <cmath>
#define HOMO_MICRO 1
__device__ __forceinline__ unsigned long long wall_clock64_() { return __builtin_amdgcn_s_memrealtime(); }

__device__ __forceinline__ bool jd_mid(double x) { return ((((unsigned)__double2hiint(x)) >> 20) & 0x7ffu) - 643u <= 760u; }
__device__ __forceinline__ double jd_div(double n, double d) {
    if (!(jd_mid(n) && jd_mid(d))) return n / d;
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = n * r;
    e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}
__device__ __forceinline__ double jd_sqrt_1to2(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
template <bool LEAN> __device__ __forceinline__ double hyp(double a, double b) {
    a = fabs(a); b = fabs(b);
    if (LEAN) {
        if (a > b) { b = jd_div(b, a); return a * jd_sqrt_1to2(1 + b * b); }
        if (b > 0) { a = jd_div(a, b); return b * jd_sqrt_1to2(1 + a * a); }
    } else {
        if (a > b) { b /= a; return a * sqrt(1 + b * b); }
        if (b > 0) { a /= b; return b * sqrt(1 + a * a); }
    }
    return 0;
}

struct Stamp { unsigned long long c0, c1, r0, r1; };
#define BEGIN const unsigned long long r0 = wall_clock64_(); const unsigned long long c0 = __builtin_readcyclecounter();
#define END(o) const unsigned long long c1 = __builtin_readcyclecounter(); const unsigned long long r1 = wall_clock64_(); if (threadIdx.x == 0) { st->c0 = c0; st->c1 = c1; st->r0 = r0; st->r1 = r1; } out[threadIdx.x] = (o);

__global__ void k_fma(Stamp* st, double* out, double x, int n) {
    double a = x + threadIdx.x;
    BEGIN
    for (int i = 0; i < n; i++) a = __builtin_fma(a, 0.999, 0.5);
    END(a)
}
__global__ void k_div(Stamp* st, double* out, double x, int n) {
    double a = x + threadIdx.x;
    BEGIN
    for (int i = 0; i < n; i++) a = 3.0 / a + 1.0;
    END(a)
}
__global__ void k_divlean(Stamp* st, double* out, double x, int n) {
    double a = x + threadIdx.x;
    BEGIN
    for (int i = 0; i < n; i++) a = jd_div(3.0, a) + 1.0;
    END(a)
}
__global__ void k_sqrt(Stamp* st, double* out, double x, int n) {
    double a = x + threadIdx.x;
    BEGIN
    for (int i = 0; i < n; i++) a = sqrt(a) + 1.0;
    END(a)
}
template <bool LEAN> __global__ void k_math(Stamp* st, double* out, double x, int n) {
    double p = x + threadIdx.x * 0.001, Wk = 1.0, Wl = 2.5;
    BEGIN
    for (int i = 0; i < n; i++) {
        const double y = (Wl - Wk) * 0.5;
        double tt = fabs(y) + hyp<LEAN>(p, y);
        double sn = hyp<LEAN>(p, tt);
        double c, s2, t2;
        if (LEAN) { c = jd_div(tt, sn); s2 = jd_div(p, sn); t2 = jd_div(p, tt) * p; }
        else { c = tt / sn; s2 = p / sn; t2 = (p / tt) * p; }
        Wk -= t2 * 1e-3; Wl += t2 * 1e-3; p = p * c + s2 * 1e-3;
    }
    END(p + Wk + Wl)
}
struct JCand { double v; int ord, a, b; };
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ JCand jc_dpp(const JCand& c) {
    JCand r;
    const int chi = __double2hiint(c.v), clo = __double2loint(c.v), cpk = (c.ord << 16) | (c.a << 8) | c.b;
    const int hi = __builtin_amdgcn_update_dpp(chi, chi, CTRL, ROW_MASK, 0xf, false);
    const int lo = __builtin_amdgcn_update_dpp(clo, clo, CTRL, ROW_MASK, 0xf, false);
    const int pk = __builtin_amdgcn_update_dpp(cpk, cpk, CTRL, ROW_MASK, 0xf, false);
    r.v = __hiloint2double(hi, lo);
    r.ord = pk >> 16; r.a = (pk >> 8) & 0xff; r.b = pk & 0xff;
    return r;
}
__device__ __forceinline__ JCand jc_best(const JCand& x, const JCand& y) {
    const unsigned long long ax = (unsigned long long)__double_as_longlong(x.v) & 0x7fffffffffffffffull;
    const unsigned long long ay = (unsigned long long)__double_as_longlong(y.v) & 0x7fffffffffffffffull;
    const bool take_y = ay > ax || (ay == ax && y.ord < x.ord);
    JCand r;
    r.v = take_y ? y.v : x.v; r.ord = take_y ? y.ord : x.ord; r.a = take_y ? y.a : x.a; r.b = take_y ? y.b : x.b;
    return r;
}
__device__ __forceinline__ JCand jc_wave_argmax(JCand c) {
    c = jc_best(c, jc_dpp<0xB1>(c));
    c = jc_best(c, jc_dpp<0x4E>(c));
    c = jc_best(c, jc_dpp<0x141>(c));
    c = jc_best(c, jc_dpp<0x140>(c));
    c = jc_best(c, jc_dpp<0x142, 0xa>(c));
    c = jc_best(c, jc_dpp<0x143, 0xc>(c));
    return c;
}
__global__ void k_argmax(Stamp* st, double* out, double x, int n) {
    JCand c; c.v = x * (threadIdx.x * 37 % 64); c.ord = threadIdx.x; c.a = threadIdx.x & 15; c.b = threadIdx.x >> 4;
    double acc = 0;
    BEGIN
    for (int i = 0; i < n; i++) {
        JCand w = jc_wave_argmax(c);
        const int k = __builtin_amdgcn_readlane(w.a, 63);
        const double p = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(w.v), 63), __builtin_amdgcn_readlane(__double2loint(w.v), 63));
        acc += p;
        c.v = c.v * 0.99 + (threadIdx.x == k ? p * 0.01 : 0.0);
    }
    END(acc)
}
__global__ void k_lds(Stamp* st, double* out, double x, int n) {
    __shared__ double A[128];
    A[threadIdx.x] = x + threadIdx.x; A[threadIdx.x + 64] = x;
    __syncthreads();
    int j = threadIdx.x;
    double acc = 0;
    BEGIN
    for (int i = 0; i < n; i++) {
        const double v = A[j];
        acc += v;
        j = ((int)v + i) & 63;
        A[(j + 7) & 63] = acc;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    END(acc)
}


// ---- the forms the library uses now ----
__device__ __forceinline__ double jd_div2(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = n * r;
    e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}
__device__ __forceinline__ bool jd_mid2(int hi_word) { return (((unsigned)hi_word >> 20) & 0x7ffu) - 653u <= 740u; }
__global__ void k_math2(Stamp* st, double* out, double x, int n) {
    double p = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x))), Wk = 1.0, Wl = 2.5;
    BEGIN
    for (int i = 0; i < n; i++) {
        const double y = (Wl - Wk) * 0.5;
        double tt, sn, c;
        if (jd_mid2(__builtin_amdgcn_readfirstlane(__double2hiint(p))) && jd_mid2(__builtin_amdgcn_readfirstlane(__double2hiint(y)))) {
            const double ap = fabs(p), ay = fabs(y);
            const double big = fmax(ap, ay), q1 = jd_div2(fmin(ap, ay), big);
            tt = ay + big * jd_sqrt_1to2(1 + q1 * q1);
            const double q2 = jd_div2(ap, tt);
            sn = tt * jd_sqrt_1to2(1 + q2 * q2);
            c = jd_div2(tt, sn);
            sn = jd_div2(p, sn); tt = jd_div2(p, tt) * p;
        } else {
            tt = fabs(y) + hyp<false>(p, y);
            sn = hyp<false>(p, tt);
            c = tt / sn; sn = p / sn; tt = (p / tt) * p;
        }
        if (y < 0) sn = -sn, tt = -tt;
        Wk -= tt * 1e-3; Wl += tt * 1e-3; p = p * c + sn * 1e-3;
    }
    END(p + Wk + Wl)
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double jc_max_step(double v) {
    const int hi = __double2hiint(v), lo = __double2loint(v);
    const int ohi = ROW_MASK == 0xf ? 0 : hi, olo = ROW_MASK == 0xf ? 0 : lo;
    const double o = __hiloint2double(__builtin_amdgcn_update_dpp(ohi, hi, CTRL, ROW_MASK, 0xf, false), __builtin_amdgcn_update_dpp(olo, lo, CTRL, ROW_MASK, 0xf, false));
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(o));
    return r;
}
__device__ __forceinline__ double jc_row_max(double v) {
    v = jc_max_step<0xB1>(v); v = jc_max_step<0x4E>(v); v = jc_max_step<0x141>(v); v = jc_max_step<0x140>(v);
    return v;
}
__device__ __forceinline__ double rl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__global__ void k_pivot2(Stamp* st, double* out, double x, int n) {      // max-reduce over 32 lanes + ballot + ff1 + 3 readlanes
    double v = x * (threadIdx.x * 37 % 64);
    double acc = 0;
    BEGIN
    for (int i = 0; i < n; i++) {
        const double key = threadIdx.x < 32 ? fabs(v) : -1.;
        const double m32 = jc_max_step<0x142, 0x2>(jc_row_max(key));
        const double mx = rl(m32, 31);
        const unsigned long long eq = __ballot(key == mx);
        const int L = __builtin_ctzll(eq | (1ull << 63));
        const int wk = __builtin_amdgcn_readlane((int)threadIdx.x * 3, L);
        const double p = rl(v, L);
        acc += p + wk;
        v = v * 0.99 + (threadIdx.x == (wk & 63) ? p * 0.01 : 0.0);
    }
    END(acc)
}
__global__ void k_rescan2(Stamp* st, double* out, double x, int n) {     // four in-row max-reductions + ballot + 8 readlanes + selects
    double v = x * (threadIdx.x * 37 % 64);
    double acc = 0;
    const int q = threadIdx.x & 15;
    BEGIN
    for (int i = 0; i < n; i++) {
        const double key = fabs(v);
        const double mx = jc_row_max(key);
        const unsigned long long eq = __ballot(key == mx);
        double wv[4];
        int w[4];
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            const unsigned bits = (unsigned)(eq >> (16 * gg)) & 0xffffu;
            w[gg] = __builtin_ctz(bits | 0x10000u) & 15;
            wv[gg] = rl(v, 16 * gg + w[gg]);
        }
        if (q == w[0]) v = v * 0.5 + wv[1] * 0.1;
        if (q == w[2]) v = v * 0.5 + wv[3] * 0.1;
        acc += wv[0] + wv[2];
    }
    END(acc)
}

// round 4: the accumulating wave of homography.hip's ordered_sums -- one 16-byte slot per point (pitch 45 slots), two dependent adds
__global__ void k_accum(Stamp* st, double* out, double x, int n) {
    __shared__ double2 P[16 * 45];
    for (int i = threadIdx.x; i < 16 * 45; i += blockDim.x) P[i] = make_double2(x * 1e-9 + i * 1e-12, x * 1e-9 - i * 1e-12);
    __syncthreads();
    double acc = x;
    const double2* p = P + (threadIdx.x % 45);
    BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u0 = 0; u0 < 16; u0 += 8) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = p[(u0 + u) * 45];
#pragma unroll
            for (int u = 0; u < 8; u++) { acc += v[u].x; acc += v[u].y; }
        }
        asm volatile("" ::: "memory");
    }
    END(acc)
}
__global__ void k_add(Stamp* st, double* out, double x, int n) {
    double a = x + threadIdx.x;
    BEGIN
    for (int i = 0; i < n; i++) { a += 0.5; a += 0.25; a += 0.125; a += 0.0625; }
    END(a)
}
__global__ void k_addf(Stamp* st, double* out, double x, int n) {
    float a = (float)x + threadIdx.x;
    BEGIN
    for (int i = 0; i < n; i++) { a += 0.5f; a += 0.25f; a += 0.125f; a += 0.0625f; }
    END((double)a)
}

// throughput of independent f64 operations on one wave: 8 chains
template <int OP> __global__ void k_tput(Stamp* st, double* out, double x, int n) {
    double a[8];
    for (int i = 0; i < 8; i++) a[i] = x + threadIdx.x + i;
    BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (OP == 0) a[j] = a[j] + 0.5;
            else if (OP == 1) a[j] = a[j] * 0.999;
            else a[j] = __builtin_fma(a[j], 0.999, 0.5);
        }
    }
    END(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7])
}
__global__ void k_mul_dep(Stamp* st, double* out, double x, int n) {
    double a = x + threadIdx.x;
    BEGIN
    for (int i = 0; i < n; i++) { a *= 0.999; a *= 1.001; a *= 0.999; a *= 1.001; }
    END(a)
}

template <typename K> void run(const char* name, K kern, int n, int blocks, int threads = 64) {
    Stamp* st; double* out;
    hipMalloc(&st, sizeof(Stamp)); hipMalloc(&out, 8 * 1024);
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, st, out, 1.5, n); hipDeviceSynchronize(); }
    Stamp h; hipMemcpy(&h, st, sizeof(h), hipMemcpyDeviceToHost);
    const double cyc = double(h.c1 - h.c0), us = double(h.r1 - h.r0) * 0.01;
    printf("%-26s blocks %4d  n %6d: %8.1f cycles / iteration  %8.4f us / iteration  (counter %.0f MHz)\n", name, blocks, n, cyc / n, us / n, cyc / us);
    hipFree(st); hipFree(out);
}
int main() {
    for (int blocks : {1, 120}) {
        run("dependent f64 fma", k_fma, 4000, blocks);
        run("4 dependent f64 adds", k_add, 4000, blocks);
        run("4 dependent f32 adds", k_addf, 4000, blocks);
        run("accumulate 16 slots", k_accum, 1000, blocks);
        run("4 dependent f64 muls", k_mul_dep, 4000, blocks);
        run("8 independent f64 adds", k_tput<0>, 4000, blocks);
        run("8 independent f64 muls", k_tput<1>, 4000, blocks);
        run("8 independent f64 fmas", k_tput<2>, 4000, blocks);
        run("... f64 muls, 2 waves / block", k_tput<1>, 4000, blocks, 128);
        run("... f64 muls, 4 waves / block", k_tput<1>, 4000, blocks, 256);
        run("... f64 muls, 8 waves / block", k_tput<1>, 4000, blocks, 512);
        run("... f64 adds, 4 waves / block", k_tput<0>, 4000, blocks, 256);
        run("f64 div (generic) + add", k_div, 1000, blocks);
        run("f64 div (lean) + add", k_divlean, 1000, blocks);
        run("f64 sqrt (generic) + add", k_sqrt, 1000, blocks);
        run("rotation math generic", k_math<false>, 500, blocks);
        run("rotation math lean", k_math<true>, 500, blocks);
        run("rotation math, one guard", k_math2, 500, blocks);
        run("wave argmax + readlanes", k_argmax, 500, blocks);
        run("pivot: max32 + ballot + ff1", k_pivot2, 500, blocks);
        run("4 row scans: max16 + ballot", k_rescan2, 500, blocks);
        run("lds round trip chain", k_lds, 1000, blocks);
    }
    return 0;
}
