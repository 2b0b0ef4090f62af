// Accuracy of the fp64 seed instructions v_rsq_f64 / v_rcp_f64 / v_sqrt_f64 on this chip (max relative error over random
// arguments), to size the Newton refinements of rsqrt_pivot (chol.hip) and fast_rcp (matern_device.hpp):
//   hipcc --offload-arch=gfx950 -O2 tools/diag/seed_precision.hip -o /tmp/seed_precision && /tmp/seed_precision
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void seeds(const double *x, double *rsq, double *rcp, double *sq, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rsq[i] = __builtin_amdgcn_rsq(x[i]);
    rcp[i] = __builtin_amdgcn_rcp(x[i]);
    sq[i] = __builtin_amdgcn_sqrt(x[i]);
}

int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), a(n), b(n), c(n);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> mant(1.0, 4.0);
    std::uniform_int_distribution<int> ex(-40, 40);
    for (int i = 0; i < n; ++i) x[i] = std::ldexp(mant(g), ex(g));
    double *dx, *da, *db, *dc;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(seeds, dim3(n / 256), dim3(256), 0, 0, dx, da, db, dc, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 8, hipMemcpyDeviceToHost);
    long double e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        long double xi = x[i];
        long double t1 = 1.0L / sqrtl(xi), t2 = 1.0L / xi, t3 = sqrtl(xi);
        e1 = fmaxl(e1, fabsl((a[i] - t1) / t1));
        e2 = fmaxl(e2, fabsl((b[i] - t2) / t2));
        e3 = fmaxl(e3, fabsl((c[i] - t3) / t3));
    }
    printf("max relative error over %d arguments: v_rsq_f64 %.3Le (2^%.1Lf)  v_rcp_f64 %.3Le (2^%.1Lf)  v_sqrt_f64 %.3Le (2^%.1Lf)\n", n, e1,
           log2l(e1), e2, log2l(e2), e3, log2l(e3));
    return 0;
}
