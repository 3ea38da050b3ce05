// tools/gap_probe.hip — where do the 6-7 us gaps between dependent kernels of one stream come from? (diagnostic, not part of libxqhip)
//
// Stream A runs pairs P -> S; device timestamps (s_memrealtime, 100 MHz) give gap = start(S) - end(P).  Variants of P (how much it
// writes, how long it runs), of S (LDS size, grid, pointer arguments into signal memory) and of what the OTHER streams are doing
// (idle / blocked in a wait / running kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d: %s\n", hipGetErrorString(e_), __LINE__, #x); exit(1); } } while (0)

struct Stamps { unsigned long long start, end; };

__global__ void work(float* buf, long long floats_per_block, int spin, Stamps* st, unsigned* sigptr, int store_mode, unsigned long long* ends) {
    extern __shared__ float lds[];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st->start = __builtin_amdgcn_s_memrealtime();
        if (sigptr) __hip_atomic_store(sigptr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    float v = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    float* mine = buf + (long long)blockIdx.x * floats_per_block;
    for (long long i = threadIdx.x; i < floats_per_block; i += blockDim.x) {
        if (store_mode == 1) __builtin_nontemporal_store(v, mine + i);
        else if (store_mode == 2) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        else if (store_mode == 3) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else mine[i] = v;
    }
    if (v == 12345.f) lds[0] = v;
    __syncthreads();
    if (threadIdx.x == 0) ends[blockIdx.x] = __builtin_amdgcn_s_memrealtime();      // one slot per block: no same-address atomics
}
__global__ void spin_only(float* p, int spin) {
    float v = p[threadIdx.x];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}

int main() {
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    const size_t big = 64u << 20;
    float* buf; CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 0, big));
    float* buf_uc = nullptr; float* buf_fg = nullptr;
    if (hipExtMallocWithFlags((void**)&buf_uc, big, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); buf_uc = nullptr; printf("no uncached alloc\n"); }
    if (hipExtMallocWithFlags((void**)&buf_fg, big, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); buf_fg = nullptr; printf("no fine-grained alloc\n"); }
    float* small; CK(hipMalloc(&small, 1 << 20)); CK(hipMemset(small, 0, 1 << 20));
    unsigned* sig; CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); CK(hipMemset(sig, 0, 8));
    unsigned* sig2; CK(hipExtMallocWithFlags((void**)&sig2, 8, hipMallocSignalMemory)); CK(hipMemset(sig2, 0, 8));
    const int reps = 100;
    Stamps* st; CK(hipMalloc(&st, sizeof(Stamps) * 2 * reps));
    const int kMaxGrid = 4096;
    unsigned long long* ends; CK(hipMalloc(&ends, sizeof(unsigned long long) * kMaxGrid * 2 * reps));
    CK(hipFuncSetAttribute((const void*)work, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipEvent_t never; CK(hipEventCreateWithFlags(&never, hipEventDisableTiming));
    struct Var { const char* name; long long p_bytes; int p_spin; int p_grid; int s_lds; int s_grid; int s_sigarg; int other; int store_mode; int alloc; };
    // other: 0 idle, 1 B blocked in hipStreamWaitValue32 (released at the end), 2 B runs spin kernels concurrently
    const Var vars[] = {
        {"tiny P -> S 720 blocks, no LDS", 0, 300, 256, 0, 720, 0, 0, 0, 0},
        {"tiny P -> S 720 blocks, 16 KB LDS", 0, 300, 256, 16 * 1024, 720, 0, 0, 0, 0},
        {"tiny P -> S 720 blocks, 32 KB LDS", 0, 300, 256, 32 * 1024, 720, 0, 0, 0, 0},
        {"tiny P -> S 720 blocks, 60 KB LDS", 0, 300, 256, 60 * 1024, 720, 0, 0, 0, 0},
        {"tiny P -> S 512 blocks, 60 KB LDS", 0, 300, 256, 60 * 1024, 512, 0, 0, 0, 0},
        {"tiny P -> S 1440 blocks, 30 KB LDS", 0, 300, 256, 30 * 1024, 1440, 0, 0, 0, 0},
    };
    for (const Var& v : vars) {
        CK(hipMemset(st, 0, sizeof(Stamps) * 2 * reps));
        CK(hipMemset(ends, 0, sizeof(unsigned long long) * kMaxGrid * 2 * reps));
        CK(hipMemset(sig, 0, 8)); CK(hipMemset(sig2, 0, 8));
        CK(hipDeviceSynchronize());
        if (v.other == 1) CK(hipStreamWaitValue32(B, sig2, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu));
        for (int r = 0; r < reps; ++r) {
            if (v.other == 2) for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin_only, dim3(128), dim3(256), 0, B, small + 4096, 2000);
            float* pb = v.alloc == 1 ? buf_uc : v.alloc == 2 ? buf_fg : buf;
            if (!pb) break;
            hipLaunchKernelGGL(work, dim3(v.p_grid), dim3(256), 0, A, pb, (long long)(v.p_bytes / 4 / v.p_grid), v.p_spin, st + 2 * r, (unsigned*)nullptr, v.store_mode, ends + (size_t)(2 * r) * kMaxGrid);
            hipLaunchKernelGGL(work, dim3(v.s_grid), dim3(256), (size_t)v.s_lds, A, small, 0LL, 300, st + 2 * r + 1, v.s_sigarg ? sig : (unsigned*)nullptr, 0, ends + (size_t)(2 * r + 1) * kMaxGrid);
        }
        CK(hipStreamSynchronize(A));
        if (v.other == 1) { unsigned one = 1; CK(hipMemcpy(sig2, &one, 4, hipMemcpyHostToDevice)); }
        CK(hipDeviceSynchronize());
        std::vector<Stamps> h(2 * reps);
        CK(hipMemcpy(h.data(), st, sizeof(Stamps) * 2 * reps, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> he((size_t)kMaxGrid * 2 * reps);
        CK(hipMemcpy(he.data(), ends, sizeof(unsigned long long) * he.size(), hipMemcpyDeviceToHost));
        for (int k = 0; k < 2 * reps; ++k) { unsigned long long m = 0; for (int b = 0; b < kMaxGrid; ++b) m = std::max(m, he[(size_t)k * kMaxGrid + b]); h[k].end = m; }
        std::vector<double> gaps, gaps2, pd, sd;
        for (int r = 5; r < reps; ++r) {
            gaps.push_back(((double)h[2 * r + 1].start - (double)h[2 * r].end) * 0.01);
            gaps2.push_back(((double)h[2 * r].start - (double)h[2 * r - 1].end) * 0.01);
            pd.push_back(((double)h[2 * r].end - (double)h[2 * r].start) * 0.01);
            sd.push_back(((double)h[2 * r + 1].end - (double)h[2 * r + 1].start) * 0.01);
        }
        std::sort(gaps.begin(), gaps.end()); std::sort(gaps2.begin(), gaps2.end()); std::sort(pd.begin(), pd.end()); std::sort(sd.begin(), sd.end());
        printf("   S runs %6.1f us   ", sd[sd.size() / 2]);
        printf("%-58s gap P->S median %6.2f us (max %6.2f)   S->next P %6.2f us   P runs %6.1f us\n", v.name, gaps[gaps.size() / 2], gaps.back(),
               gaps2[gaps2.size() / 2], pd[pd.size() / 2]);
    }
    return 0;
}
