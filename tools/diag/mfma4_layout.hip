// Determines the operand / result lane mapping of v_mfma_f64_4x4x4_4b_f64 empirically (one-hot inputs).
// build: hipcc --offload-arch=gfx950 -O2 -o mfma4_layout mfma4_layout.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k(double *out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[((size_t)la * 64 + lb) * 64 + lane] = d;
        }
}

int main()
{
    double *d;
    hipMalloc(&d, 64 * 64 * 64 * sizeof(double));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    std::vector<double> h(64 * 64 * 64);
    hipMemcpy(h.data(), d, h.size() * sizeof(double), hipMemcpyDeviceToHost);
    // for every (la, lb) with a non-zero result print the output lane(s)
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int l = 0; l < 64; ++l)
                if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" (B%d->D%d)", lb, l);
        printf("\n");
    }
    return 0;
}
