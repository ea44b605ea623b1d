// Which hardware queue does the n-th stream a process creates land on?  One empty kernel per stream; read the answer from
// `rocprofv3 --kernel-trace` (Stream_Id / Queue_Id).   hipcc --offload-arch=gfx950 -o queue_map queue_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void tick(int *p) { if (p) *p = 1; }
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;      // 0: non-blocking streams, 1: default flags, 2: alternate priorities
    hipStream_t st[32];
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    printf("priority range %d .. %d\n", lo, hi);
    for (int i = 0; i < n; ++i) {
        if (mode == 0) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        else if (mode == 1) hipStreamCreate(&st[i]);
        else hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, (i & 1) ? hi : 0);
    }
    for (int r = 0; r < 2; ++r)
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tick, dim3(1), dim3(64), 0, st[i], nullptr);
    hipLaunchKernelGGL(tick, dim3(1), dim3(64), 0, 0, nullptr);
    hipDeviceSynchronize();
    for (int i = 0; i < n; ++i) hipStreamDestroy(st[i]);
    return 0;
}
