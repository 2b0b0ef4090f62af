// diagnostic: does hipExtStreamCreateWithCUMask restrict where workgroups land on gfx950?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | (31 << 11));
        unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11));
        out[blockIdx.x] = (xcc << 24) | (hwid & 0xffffff);
    }
    // burn a little time so blocks spread
    for (volatile int i = 0; i < 2000; ++i) {}
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int ncu = p.multiProcessorCount; printf("CUs %d\n", ncu);
    const int nb = 8192;
    unsigned *d; hipMalloc(&d, nb * 4); std::vector<unsigned> h(nb);
    for (int reserve : {0, 8, 64}) {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int c = reserve; c < ncu; ++c) mask[c / 32] |= 1u << (c % 32);
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, mask.size(), mask.data());
        printf("reserve %d: create -> %s\n", reserve, hipGetErrorString(e));
        if (e != hipSuccess) continue;
        hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
        std::set<unsigned> cus; std::set<unsigned> xccs;
        for (auto v : h) { unsigned cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 0x7, xcc = v >> 24; cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc); }
        printf("  distinct CUs used: %zu, XCCs: %zu\n", cus.size(), xccs.size());
        if (reserve == 8) { // which CUs missing per xcc
            int per[8] = {0}; for (auto c : cus) per[c >> 16]++; for (int x = 0; x < 8; ++x) printf("  xcc %d: %d CUs\n", x, per[x]);
        }
        hipStreamDestroy(s);
    }
    return 0;
}
