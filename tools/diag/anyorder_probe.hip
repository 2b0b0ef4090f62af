// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950 (hip_ext.h says "not supported on GFX9xx")?
// Kernel A spins (bounded: 50 ms of the 100 MHz clock) until kernel B, launched BEHIND it on the same stream, raises a
// flag.  Prints the ticks A waited in three launch modes.   hipcc --offload-arch=gfx950 -O2 anyorder_probe.hip -o anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>

__global__ void waiter(unsigned *flag, unsigned long long *out)
{
    if (threadIdx.x) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dt = 0;
    for (;;) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        dt = __builtin_amdgcn_s_memrealtime() - t0;
        if (dt > 5000000ull) break;
        __builtin_amdgcn_s_sleep(16);
    }
    out[0] = __builtin_amdgcn_s_memrealtime() - t0;
    out[1] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void raiser(unsigned *flag)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main()
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *flag; unsigned long long *out, h[2];
    hipMalloc(&flag, 4); hipMalloc(&out, 16);
    for (int mode = 0; mode < 3; ++mode) {
        hipMemsetAsync(flag, 0, 4, s);
        hipStreamSynchronize(s);
        const int fa = mode == 2 ? hipExtAnyOrderLaunch : 0, fb = mode >= 1 ? hipExtAnyOrderLaunch : 0;
        hipExtLaunchKernelGGL(waiter, dim3(1), dim3(64), 0, s, nullptr, nullptr, fa, flag, out);
        hipExtLaunchKernelGGL(raiser, dim3(1), dim3(64), 0, s, nullptr, nullptr, fb, flag);
        hipStreamSynchronize(s);
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("mode %d (A %s, B %s): A waited %.1f us, flag seen %llu  -> %s\n", mode, fa ? "anyorder" : "ordered",
               fb ? "anyorder" : "ordered", h[0] * 0.01, h[1], h[1] ? "kernels OVERLAPPED" : "B did not start before A ended");
    }
    printf("last error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
