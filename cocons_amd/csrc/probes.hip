// probes.hip -- measurement probes of the fp64 pipes (what v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 / v_fma_f64 sustain,
// whether the matrix and the vector pipe run side by side).  NOT part of the product: built into libcocons_hip_probes.so
// (include/cocons_hip_probes.h), which nothing but tools/probe_mfma*.py and tools/diag/corun.py loads; libcocons_hip.so
// contains none of these kernels.  The numbers they gave are in profiles/r02_mfma_f64_probe.json and DESIGN.md section 8.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/cocons_hip_probes.h"

namespace cocons {
typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
}

static thread_local std::string g_probe_err;
extern "C" const char *cocons_probe_last_error(void) { return g_probe_err.c_str(); }
static int fail(int code, const char *fmt, const char *what = "")
{
    char buf[512];
    snprintf(buf, sizeof buf, fmt, what);
    g_probe_err = buf;
    return code;
}
#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) return fail(-100 - (int)e__, "HIP: %s", hipGetErrorString(e__)); \
    } while (0)

// Measurement probe: back-to-back v_mfma_f64_16x16x4_f64 on every SIMD (one wave per
// SIMD, 4 independent accumulators, operands in registers).  Gives the fp64 matrix rate
// this chip actually sustains, the ceiling the update kernel is priced against.
namespace cocons {
__global__ void __launch_bounds__(256)
mfma_f64_probe_kernel(double *out, int iters, double seed)
{
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int i = 0; i < iters; ++i) {
        acc0 = MFMA64(a, b, acc0);
        acc1 = MFMA64(b, a, acc1);
        acc2 = MFMA64(a, a, acc2);
        acc3 = MFMA64(b, b, acc3);
    }
    d4 s = acc0 + acc1 + acc2 + acc3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// companion probe: independent v_fma_f64 chains (16 accumulators per lane), the fp64 VECTOR rate
__global__ void __launch_bounds__(256)
vfma_f64_probe_kernel(double *out, int iters, double seed)
{
    double a = seed + threadIdx.x * 1e-9, b = 1.0 - 1e-9;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fma(x[i], b, a);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Extended probe (settles what bounds the fp64 matrix pipe): NACC independent accumulators per wave,
// either the 16x16x4 form (2048 flop) or the 4x4x4 four-block form (512 flop), with in-kernel stamps:
// stamp[2 b] = shader cycles (s_memtime), stamp[2 b + 1] = ticks of the constant 100 MHz clock
// (s_memrealtime) spent by workgroup b in the loop -- their ratio is the clock the chip actually held.
template <int NACC, int FORM>
__global__ void __launch_bounds__(256)
mfma_f64_probe_ex_kernel(double *out, unsigned long long *stamp, int iters, double seed)
{
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double res = 0.0;
    if (FORM == 0) {
        d4 acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = MFMA64((i & 1) ? a : b, (i & 2) ? a : b, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[NACC];
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64((i & 1) ? a : b, (i & 2) ? a : b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) res += acc[i];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (threadIdx.x == 0) { stamp[2 * blockIdx.x] = c1 - c0; stamp[2 * blockIdx.x + 1] = r1 - r0; }
}

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)
// FORM 3: sixteen accumulators fed from four + four DISTINCT operand registers (no LDS traffic): tells
// operand-register switching apart from the LDS feed
__global__ void __launch_bounds__(256)
mfma4_regs_probe_kernel(double *out, unsigned long long *stamp, int iters, double seed)
{
    double pr[2][2], pc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int s = 0; s < 2; ++s) { pr[x][s] = seed + threadIdx.x * 1e-3 + x + 2 * s; pc[x][s] = seed - threadIdx.x * 1e-3 - x - 2 * s; }
    double acc[2][2][2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[x][y][s][t] = 0.0;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < 2 * iters; ++it) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(pr[x][s]), "+v"(pc[x][s]));
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[x][y][s][t] = MFMA4(pc[y][t], pr[x][s], acc[x][y][s][t]);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double res = 0.0;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) res += acc[x][y][s][t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if (threadIdx.x == 0) { stamp[2 * blockIdx.x] = c1 - c0; stamp[2 * blockIdx.x + 1] = r1 - r0; }
}

// idle filler: one wave that sleeps for `ticks` of the 100 MHz clock (a gap between bursts on the stream)
__global__ void idle_kernel(unsigned long long ticks)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) __builtin_amdgcn_s_sleep(64);
}

template <int NACC, int FORM>
static void launch_probe_ex(int blocks, double *dbuf, unsigned long long *stamp, int iters, hipStream_t s)
{
    hipLaunchKernelGGL((mfma_f64_probe_ex_kernel<NACC, FORM>), dim3(blocks), dim3(256), 0, s, dbuf, stamp, iters, 1.0);
}

// bursts of `iters` loop iterations separated by idle gaps of gap_us (0 = back to back), `reps` bursts
// after a warm-up of the same pattern.  out[0] = TFLOP/s inside the bursts (event-timed, gaps excluded),
// out[1] = in-kernel clock in GHz (median over workgroups of the last burst), out[2] = shader cycles per
// MFMA instruction per wave, out[3] = mean burst duration in ms
int run_mfma_f64_probe_ex(int blocks, int nacc, int form, int iters, int gap_us, int reps, double *out)
{
    hipStream_t s;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return -1;
    double *dbuf = nullptr;
    unsigned long long *dst = nullptr;
    hipMalloc(&dbuf, (size_t)blocks * 256 * sizeof(double));
    hipMalloc(&dst, (size_t)blocks * 2 * sizeof(unsigned long long));
    auto burst = [&]() {
        if (form == 3) {     // nacc is fixed at 16, iters counts groups of 32 instructions
            hipLaunchKernelGGL(mfma4_regs_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, dst, iters, 1.0);
        } else if (form == 0) {
            if (nacc == 4) launch_probe_ex<4, 0>(blocks, dbuf, dst, iters, s);
            else if (nacc == 8) launch_probe_ex<8, 0>(blocks, dbuf, dst, iters, s);
            else launch_probe_ex<16, 0>(blocks, dbuf, dst, iters, s);
        } else {
            if (nacc == 4) launch_probe_ex<4, 1>(blocks, dbuf, dst, iters, s);
            else if (nacc == 8) launch_probe_ex<8, 1>(blocks, dbuf, dst, iters, s);
            else launch_probe_ex<16, 1>(blocks, dbuf, dst, iters, s);
        }
    };
    const int warm = reps;
    std::vector<hipEvent_t> ev(2 * (size_t)reps);
    for (auto &e : ev) hipEventCreate(&e);
    for (int r = 0; r < warm + reps; ++r) {
        if (r >= warm) hipEventRecord(ev[2 * (r - warm)], s);
        burst();
        if (r >= warm) hipEventRecord(ev[2 * (r - warm) + 1], s);
        if (gap_us > 0) hipLaunchKernelGGL(idle_kernel, dim3(1), dim3(64), 0, s, (unsigned long long)gap_us * 100ull);
    }
    hipStreamSynchronize(s);
    double ms_sum = 0;
    for (int r = 0; r < reps; ++r) {
        float ms = 0;
        hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]);
        ms_sum += ms;
    }
    for (auto &e : ev) hipEventDestroy(e);
    std::vector<unsigned long long> st((size_t)blocks * 2);
    hipMemcpy(st.data(), dst, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> clk(blocks), cyc(blocks);
    for (int b = 0; b < blocks; ++b) {
        clk[b] = (double)st[2 * b] / (double)st[2 * b + 1] * 0.1;      // cycles per 10 ns -> GHz
        cyc[b] = (double)st[2 * b] / ((double)iters * (form == 3 ? 32 : nacc));
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double flop_per = form == 0 ? 2048.0 : 512.0;
    const double flops = (double)blocks * 4.0 * (double)iters * (form == 3 ? 32 : nacc) * flop_per;
    out[0] = flops / (ms_sum / reps * 1e-3) / 1e12;
    out[1] = clk[blocks / 2];
    out[2] = cyc[blocks / 2];
    out[3] = ms_sum / reps;
    hipFree(dbuf); hipFree(dst);
    hipStreamDestroy(s);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

double run_vfma_f64_probe(hipStream_t s, int blocks, int iters, double *dbuf)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(vfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, iters / 10, 1.0);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(vfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, iters, 1.0);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    double flops = (double)blocks * 256.0 * (double)iters * 16.0 * 2.0;
    return flops / (ms * 1e-3) / 1e12;
}

// both probes at once on two streams: do the matrix and the vector fp64 pipes run concurrently?
// out[0], out[1] = TFLOP/s of the MFMA / FMA kernel while the other one is running
void run_corun_probe(int blocks_mfma, int blocks_vfma, int iters_mfma, int iters_vfma, double *dbuf, double *out)
{
    hipStream_t s0, s1;
    hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipEvent_t a0, a1, b0, b1;
    hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
    double *d0 = dbuf, *d1 = dbuf + (size_t)blocks_mfma * 256;
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks_mfma), dim3(256), 0, s0, d0, 100, 1.0);
    hipLaunchKernelGGL(vfma_f64_probe_kernel, dim3(blocks_vfma), dim3(256), 0, s1, d1, 100, 1.0);
    hipStreamSynchronize(s0); hipStreamSynchronize(s1);
    hipEventRecord(a0, s0);
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks_mfma), dim3(256), 0, s0, d0, iters_mfma, 1.0);
    hipEventRecord(a1, s0);
    hipEventRecord(b0, s1);
    hipLaunchKernelGGL(vfma_f64_probe_kernel, dim3(blocks_vfma), dim3(256), 0, s1, d1, iters_vfma, 1.0);
    hipEventRecord(b1, s1);
    hipStreamSynchronize(s0); hipStreamSynchronize(s1);
    float ma = 0, mb = 0;
    hipEventElapsedTime(&ma, a0, a1);
    hipEventElapsedTime(&mb, b0, b1);
    out[0] = (double)blocks_mfma * 4 * (double)iters_mfma * 4 * 2048.0 / (ma * 1e-3) / 1e12;
    out[1] = (double)blocks_vfma * 256.0 * (double)iters_vfma * 16.0 * 2.0 / (mb * 1e-3) / 1e12;
    out[2] = ma; out[3] = mb;
    hipEventDestroy(a0); hipEventDestroy(a1); hipEventDestroy(b0); hipEventDestroy(b1);
    hipStreamDestroy(s0); hipStreamDestroy(s1);
}

double run_mfma_f64_probe(hipStream_t s, int blocks, int iters, double *dbuf)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, iters / 10, 1.0);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(mfma_f64_probe_kernel, dim3(blocks), dim3(256), 0, s, dbuf, iters, 1.0);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    double flops = (double)blocks * 4 /*waves*/ * (double)iters * 4 /*mfma*/ * 2048.0;
    return flops / (ms * 1e-3) / 1e12;
}
}  // namespace cocons

using namespace cocons;

// ---------------------------------------------------------------------------
extern "C" int cocons_mfma_f64_probe(int blocks_per_cu, double *tflops)
{
    if (!tflops || blocks_per_cu < 1 || blocks_per_cu > 8) return fail(-1, "cocons_mfma_f64_probe: bad argument");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * blocks_per_cu;
    double *d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t)blocks * 256 * sizeof(double)));
    // blocks_per_cu > 0: MFMA probe; the vector-FMA companion is reported through cocons_vfma_f64_probe
    hipStream_t ps = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    *tflops = run_mfma_f64_probe(ps, blocks, 20000, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamDestroy(ps));
    HIPCHK(hipFree(d));
    return 0;
}

// extended probe: see run_mfma_f64_probe_ex (chol.hip) for the meaning of out[0..3]
extern "C" int cocons_mfma_f64_probe_ex(int blocks_per_cu, int nacc, int form, int iters, int gap_us, int reps, double *out4)
{
    if (!out4 || blocks_per_cu < 1 || blocks_per_cu > 8 || (nacc != 4 && nacc != 8 && nacc != 16) || form < 0 || form > 3 || form == 2 ||
        iters < 1 || reps < 1 || gap_us < 0)
        return fail(-1, "cocons_mfma_f64_probe_ex: bad argument");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    if (run_mfma_f64_probe_ex(prop.multiProcessorCount * blocks_per_cu, nacc, form, iters, gap_us, reps, out4))
        return fail(-100, "cocons_mfma_f64_probe_ex: HIP error");
    return 0;
}

extern "C" int cocons_vfma_f64_probe(int blocks_per_cu, double *tflops)
{
    if (!tflops || blocks_per_cu < 1 || blocks_per_cu > 8) return fail(-1, "cocons_vfma_f64_probe: bad argument");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * blocks_per_cu;
    double *d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t)blocks * 256 * sizeof(double)));
    hipStream_t ps = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    *tflops = run_vfma_f64_probe(ps, blocks, 20000, d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamDestroy(ps));
    HIPCHK(hipFree(d));
    return 0;
}

// diagnostic (not part of the public header): MFMA and FMA probes concurrently on two streams
extern "C" int cocons_corun_probe(int bpc_mfma, int bpc_vfma, int iters_mfma, int iters_vfma, double *out4)
{
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    int bm = prop.multiProcessorCount * bpc_mfma, bv = prop.multiProcessorCount * bpc_vfma;
    double *d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t)(bm + bv) * 256 * sizeof(double)));
    run_corun_probe(bm, bv, iters_mfma, iters_vfma, d, out4);
    HIPCHK(hipGetLastError());
    HIPCHK(hipFree(d));
    return 0;
}
