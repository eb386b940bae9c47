// prints hipDeviceAttributeIsLargeBar and whether a hipMalloc block is mapped into the process (msync probe)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstdint>
#include <cerrno>
int main() {
    int v = -1; hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar: %d (%s)\n", v, hipGetErrorString(e));
    float* p = nullptr; (void)hipMalloc((void**)&p, 4096); (void)hipMemset(p, 0, 4096); (void)hipDeviceSynchronize();
    const uintptr_t pg = (uintptr_t)p & ~(uintptr_t)4095;
    const int r = msync((void*)pg, 4096, MS_ASYNC);
    printf("msync on the allocation's page: %d (errno %d)\n", r, r ? errno : 0);
    return 0;
}
