// pcie_probe: round-trip cost of the host-Env rollout's copies on this box: pinned H2D / D2H of one env step's payload
// (295 KB at 4096 envs x 18 floats), a trivial kernel between them, stream busy-wait.   hipcc -O2 -o pcie_probe pcie_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
__global__ void touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (size_t bytes : {4096ul, 73728ul, 294912ul, 1179648ul}) {
        float *h_in, *h_out, *d; hipHostMalloc((void**)&h_in, bytes); hipHostMalloc((void**)&h_out, bytes); hipMalloc((void**)&d, bytes);
        memset(h_in, 0, bytes);
        auto spin = [&] { while (hipStreamQuery(s) == hipErrorNotReady) {} };
        double t_h2d = 0, t_d2h = 0, t_rt = 0, t_k = 0; const int reps = 200;
        for (int i = 0; i < reps + 20; ++i) {
            double a = now(); hipMemcpyAsync(d, h_in, bytes, hipMemcpyHostToDevice, s); spin(); double b = now();
            hipMemcpyAsync(h_out, d, bytes, hipMemcpyDeviceToHost, s); spin(); double c = now();
            hipLaunchKernelGGL(touch, dim3((bytes / 4 + 255) / 256), dim3(256), 0, s, d, (int)(bytes / 4)); spin(); double e = now();
            hipMemcpyAsync(d, h_in, bytes, hipMemcpyHostToDevice, s);
            hipLaunchKernelGGL(touch, dim3((bytes / 4 + 255) / 256), dim3(256), 0, s, d, (int)(bytes / 4));
            hipMemcpyAsync(h_out, d, bytes, hipMemcpyDeviceToHost, s); spin(); double f = now();
            if (i >= 20) { t_h2d += b - a; t_d2h += c - b; t_k += e - c; t_rt += f - e; }
        }
        printf("%8zu B: H2D %.1f us  D2H %.1f us  kernel %.1f us  H2D+kernel+D2H %.1f us\n", bytes, t_h2d / reps, t_d2h / reps, t_k / reps, t_rt / reps);
        hipHostFree(h_in); hipHostFree(h_out); hipFree(d);
    }
    return 0;
}
