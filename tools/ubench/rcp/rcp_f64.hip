// Accuracy of v_rcp_f64 (__builtin_amdgcn_rcp) on gfx950 with 0 / 1 / 2 Newton steps against the correctly rounded 1 / x of the host:
//   hipcc --offload-arch=gfx950 -O2 -o rcp_f64 rcp_f64.hip && ./rcp_f64
// (round 6: how many steps does ke_rcp -- 1 / |kappa| of the velocity stage -- need? csrc/ltpl_hip.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double q = __builtin_amdgcn_rcp(v);
    r0[i] = q;
    q = fma(fma(-v, q, 1.0), q, q); r1[i] = q;
    q = fma(fma(-v, q, 1.0), q, q); r2[i] = q;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> x(n), r(3 * (size_t)n);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> e(-40.0, 12.0), m(1.0, 2.0);
    for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), (int)e(g));           // |kappa| from 1e-12 to 4e3 1/m
    double *dx, *dr;
    hipMalloc(&dx, sizeof(double) * n); hipMalloc(&dr, sizeof(double) * 3 * n);
    hipMemcpy(dx, x.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, dr + n, dr + 2 * (size_t)n, n);
    hipMemcpy(r.data(), dr, sizeof(double) * 3 * n, hipMemcpyDeviceToHost);
    for (int s = 0; s < 3; ++s) {
        double worst = 0.0; long exact = 0;
        for (int i = 0; i < n; ++i) {
            const double ref = 1.0 / x[i], err = std::fabs(r[(size_t)s * n + i] - ref) / ref;
            worst = err > worst ? err : worst; exact += r[(size_t)s * n + i] == ref;
        }
        printf("v_rcp_f64 + %d Newton step(s): max relative error %.3e (%.2f ulp), correctly rounded in %.2f %% of %d samples\n", s, worst, worst / 1.11e-16, 100.0 * exact / n, n);
    }
    return 0;
}
