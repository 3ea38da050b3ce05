// tools/sync_probe.hip — what a cross-stream dependency costs the PRODUCING stream on gfx950 (diagnostic, not part of libxqhip).
//
// Stream A runs a chain of short kernels K1 K2 K3 ...; stream B must start a kernel after K1.  Variants:
//   0  no dependency at all (baseline chain time)
//   1  hipEventRecord(ev, A) after K1 + hipStreamWaitEvent(B, ev)                      (what libxqhip does)
//   2  K2 itself stores a flag at its start, B waits with hipStreamWaitValue32          (no packet on A)
// Prints the chain time on A per iteration and when B's kernel started relative to K1's end.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d: %s\n", hipGetErrorString(e_), __LINE__, #x); exit(1); } } while (0)

__global__ void spin_kernel(float* p, int iters, unsigned* flag, unsigned value, unsigned long long* stamp) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (stamp) *stamp = __builtin_amdgcn_s_memrealtime();            // 100 MHz
        if (flag) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float v = p[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}
__global__ void stamp_kernel(unsigned long long* out) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = __builtin_amdgcn_s_memrealtime(); }

int main() {
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    float* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
    unsigned* flag = nullptr;
    hipError_t fe = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
    if (fe != hipSuccess) { printf("hipMallocSignalMemory unavailable: %s\n", hipGetErrorString(fe)); flag = nullptr; }
    else CK(hipMemset(flag, 0, 8));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 2 * 200 * 8)); CK(hipMemset(stamps, 0, 2 * 200 * 8));
    hipEvent_t ev, t0, t1;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const int iters = 300, reps = 200, chain = 6;      // each kernel ~ 5 us on 256 blocks
    std::vector<unsigned long long> hs(2 * reps);
    //   3  stream A waits (hipStreamWaitValue32) between K1 and K2 for a flag stream B raised long ago      (join, already satisfied)
    //   4  stream A waits (hipStreamWaitEvent) between K1 and K2 for an event of stream B recorded long ago  (join, already satisfied)
    unsigned* flag2 = nullptr;
    if (flag) { CK(hipExtMallocWithFlags((void**)&flag2, 8, hipMallocSignalMemory)); CK(hipMemset(flag2, 0, 8)); }
    hipEvent_t evB; CK(hipEventCreateWithFlags(&evB, hipEventDisableTiming));
    //   5  K1 launched with hipExtLaunchKernelGGL(..., stopEvent = ev): the event rides on K1's own completion signal, no marker packet
    // event flags: round 5 asks whether the record's system-scope release is what costs the recording stream its ~6 us
    const unsigned evflags[3] = { hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence, hipEventDisableTiming | hipEventReleaseToDevice };
    const char* evnames[3] = { "DisableTiming", "DisableTiming|DisableSystemFence", "DisableTiming|ReleaseToDevice" };
    for (int fl = 0; fl < 3; ++fl)
    for (int variant = 0; variant < 6; ++variant) {
        if ((variant == 2 || variant == 3) && !flag) continue;
        if (fl > 0 && variant != 1 && variant != 4 && variant != 5) continue;
        if (fl > 0 && variant == 1) {
            printf("event flags: %s\n", evnames[fl]);
            CK(hipEventDestroy(ev)); CK(hipEventDestroy(evB));
            CK(hipEventCreateWithFlags(&ev, evflags[fl])); CK(hipEventCreateWithFlags(&evB, evflags[fl]));
        }
        for (int warm = 0; warm < 2; ++warm) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, A));
            for (int r = 0; r < reps; ++r) {
                const unsigned epoch = (unsigned)(variant * 100000 + warm * 1000 + r + 1);
                if (variant == 3 || variant == 4) {
                    // stream B: one short kernel per iteration, raising flag2 / recording evB behind it; A joins it one iteration later
                    hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, B, p + 512, iters, (unsigned*)nullptr, 0u, (unsigned long long*)nullptr);
                    if (variant == 3) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, B, p + 1024, 1, flag2, epoch, (unsigned long long*)nullptr);
                    else CK(hipEventRecord(evB, B));
                }
                if (variant == 5) hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, A, nullptr, ev, 0, p, iters, (unsigned*)nullptr, 0u, (unsigned long long*)nullptr);
                else hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, A, p, iters, (unsigned*)nullptr, 0u, (unsigned long long*)nullptr);   // K1
                if (variant == 3 && r > 0) CK(hipStreamWaitValue32(A, flag2, epoch - 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
                if (variant == 4 && r > 0) CK(hipStreamWaitEvent(A, evB, 0));
                if (variant == 1) { CK(hipEventRecord(ev, A)); CK(hipStreamWaitEvent(B, ev, 0)); }
                if (variant == 5) CK(hipStreamWaitEvent(B, ev, 0));
                if (variant == 2) CK(hipStreamWaitValue32(B, flag, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
                if (variant == 1 || variant == 2 || variant == 5) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(256), 0, B, p + 512, iters, (unsigned*)nullptr, 0u, stamps + 2 * r + 1);
                for (int k = 1; k < chain; ++k)     // K2 stamps its start (= K1's end) and, variant 2, raises the flag
                    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, A, p, iters, (k == 1 && variant == 2) ? flag : nullptr, epoch,
                                       k == 1 ? stamps + 2 * r : nullptr);
            }
            CK(hipEventRecord(t1, A));
            CK(hipStreamSynchronize(A));
            CK(hipStreamSynchronize(B));
            float ms = 0; CK(hipEventElapsedTime(&ms, t0, t1));
            if (warm) {
                printf("variant %d: %.2f us per iteration of %d kernels on stream A (%.2f us per kernel)", variant, 1e3f * ms / reps, chain,
                       1e3f * ms / reps / chain);
                if (variant == 1 || variant == 2 || variant == 5) {
                    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
                    double sum = 0, mx = 0;
                    for (int r = 0; r < reps; ++r) { const double d = ((double)hs[2 * r + 1] - (double)hs[2 * r]) * 0.01; sum += d; if (d > mx) mx = d; }
                    printf(";  stream B's kernel starts %.2f us (max %.2f) after the start of K2", sum / reps, mx);
                }
                printf("\n");
            }
        }
    }
    return 0;
}
