// How many workgroups of the trailing update's footprint (256 threads, 64 VGPRs, 20 KB LDS) does the chip
// hold at once -- alone, and while another stream keeps ONE resident workgroup (a "hog": 64 threads, 160 KB
// LDS, sleeping) on the device?  Every probe workgroup sleeps a fixed time, so the launch takes
// ceil(workgroups / slots) rounds; slots follow from the elapsed time, and the per-CU peak from HW_ID stamps.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/occ tools/diag/occupancy_probe.hip ; run: /tmp/occ
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256, 8) probe(unsigned long long ticks, unsigned long long *rec)
{
    __shared__ double pad[2560];                       // 20,480 B
    if (threadIdx.x == 0) pad[0] = 1.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID bits 0..3
        rec[3 * blockIdx.x] = t0;
        rec[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        rec[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
}

// mode 0: s_memrealtime + s_sleep loop; 1: s_sleep only (fixed count); 2: busy ALU loop, no sleep, no clock reads;
// 3: s_memrealtime loop without sleep; 4: polls a memory word (agent-scope atomic load) with s_sleep, like the engine
__global__ void hog(unsigned long long ticks, int mode, unsigned *word, double *sink)
{
    extern __shared__ double hs[];
    if (threadIdx.x == 0) hs[0] = 1.0;
    if (mode == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    } else if (mode == 1) {
        for (unsigned long long i = 0; i < ticks / 4; ++i) __builtin_amdgcn_s_sleep(64);     // ~64*64 cycles each
    } else if (mode == 2) {
        double x = threadIdx.x;
        for (unsigned long long i = 0; i < ticks * 4; ++i) x = x * 1.0000001 + 1e-9;
        if (x == 12345.0) sink[0] = x;
    } else if (mode == 3) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    } else {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long n = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && ++n < ticks / 8) __builtin_amdgcn_s_sleep(16);
        if (t0 == 1) sink[0] = 1.0;
    }
}

int main()
{
    hipStream_t s, h;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&h, hipStreamNonBlocking);
    const int nwg = 2048 * 4;
    unsigned long long *rec;
    hipMalloc(&rec, 3 * nwg * sizeof(unsigned long long));
    hipFuncSetAttribute((const void *)hog, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned *word; double *sink;
    hipMalloc(&word, 64); hipMemset(word, 0, 64); hipMalloc(&sink, 64);
    const int nmodes = 5;
    for (int withhog = 0; withhog < 1 + nmodes; ++withhog) {
        if (withhog) {
            printf("hog mode %d\n", withhog - 1);
            hipLaunchKernelGGL(hog, dim3(1), dim3(64), 160 * 1024, h, 300000ull, withhog - 1, word, sink);   // ~3 ms
            hipStreamSynchronize(s);
            // crude: give the hog time to become resident
            hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, s, 5000ull, rec);
            hipStreamSynchronize(s);
        }
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), 0, s, 2000ull, rec);     // 20 us each
            hipEventRecord(e1, s);
            hipStreamSynchronize(s);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> r(3 * nwg);
            hipMemcpy(r.data(), rec, r.size() * 8, hipMemcpyDeviceToHost);
            // peak concurrency overall: sweep over start/end stamps
            std::vector<std::pair<unsigned long long, int>> ev;
            for (int i = 0; i < nwg; ++i) { ev.push_back({r[3 * i], 1}); ev.push_back({r[3 * i + 1], -1}); }
            std::sort(ev.begin(), ev.end());
            int cur = 0, peak = 0;
            for (auto &x : ev) { cur += x.second; peak = std::max(peak, cur); }
            printf("hog workgroups %d rep %d: %d workgroups x 20 us took %.1f us -> %.2f rounds, peak concurrent %d\n",
                   withhog == 0 ? 0 : 1, rep, nwg, ms * 1e3, ms * 1e3 / 20.0, peak);
        }
        // placement of a single-round launch (1176 workgroups, like a mid-size trailing update): workgroups per CU
        {
            const int n1 = 1176;
            hipLaunchKernelGGL(probe, dim3(n1), dim3(256), 0, s, 2000ull, rec);
            hipStreamSynchronize(s);
            std::vector<unsigned long long> r(3 * n1);
            hipMemcpy(r.data(), rec, r.size() * 8, hipMemcpyDeviceToHost);
            std::vector<int> percu(8 * 64, 0);
            for (int i = 0; i < n1; ++i) {
                unsigned hw = (unsigned)r[3 * i + 2], xcc = (unsigned)(r[3 * i + 2] >> 32) & 15;
                unsigned cu = (hw >> 8) & 15, se = (hw >> 13) & 7;          // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
                percu[xcc * 64 + se * 16 + cu]++;
            }
            int hist[16] = {0}, used = 0;
            for (int v : percu) { if (v) ++used; hist[v < 15 ? v : 15]++; }
            printf("  single round of %d: CUs used %d; histogram of workgroups per CU:", n1, used);
            for (int v = 1; v < 12; ++v) printf(" %d:%d", v, hist[v]);
            unsigned long long tmin = ~0ull, tmax = 0;
            for (int i = 0; i < n1; ++i) { tmin = std::min(tmin, r[3 * i]); tmax = std::max(tmax, r[3 * i]); }
            printf("; start spread %.2f us\n", (tmax - tmin) * 0.01);
        }
        hipStreamSynchronize(h);
    }
    return 0;
}
