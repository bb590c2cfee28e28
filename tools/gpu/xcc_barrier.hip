// xcc_barrier.hip — microbenchmark (not product code), VERDICT r4 next #1: "price the per-XCC stage of a grid barrier at 32
// workgroups before building an XCD-local fused chain for the 64-token level; build only if < 2 us".
//
// One persistent launch, one 256-thread workgroup per CU.  Every workgroup reads its XCC id (s_getreg HW_REG_XCC_ID), a census
// pass counts the workgroups of each XCC (nothing is assumed about placement), then ITERS rounds of
//     publish a 128-byte record (round number) -> barrier among the workgroups of MY XCC only -> read the record of the next
//     workgroup of my XCC and check it
// in three publish / consume forms:
//   0  plain stores, lane-0 agent-scope RELEASE fence, relaxed arrive / poll on the XCC's counter, agent-scope ACQUIRE fence,
//      plain loads                                                   (MI355X_MICROARCH.md: the valid plain-store recipe)
//   1  sc1 (write-through) stores, s_waitcnt vmcnt(0), relaxed arrive / poll, sc1 loads, NO fences   (the granule recipe)
//   2  the barrier alone (no record published or read): arrive / poll cost at 32 workgroups per counter
// and, for scale, 3 = the same three steps with ONE counter for all 256 workgroups (form 1's data path).
// Prints microseconds per round (wall clock of the launch / ITERS; the launch itself is ~10 us of ~ITERS x several us) and the
// number of stale records seen.  Every spin is bounded (a timeout aborts the round loop and is reported).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

struct Ctl {
    unsigned census[8];       // workgroups per XCC
    unsigned rank_next[8];    // census tickets
    unsigned all_in;          // census barrier (all workgroups)
    unsigned timeout;         // set when a spin gave up
    unsigned stale;           // records that did not carry the expected round
    unsigned pad[13];
    unsigned cnt[8][32];      // per-XCC barrier counters, one cache line each
    unsigned cnt_all[32];     // one counter for everybody (form 3)
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target, unsigned* timeout) {
    for (int it = 0; it < (1 << 22); ++it) {
        if ((int)(ld_relaxed(p) - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    atomicExch(timeout, 1u);
    return false;
}

__global__ __launch_bounds__(256) void xcc_barrier_kernel(Ctl* ctl, unsigned* rec /* [blocks][32] */, unsigned* slot_of /* [8][256] */,
                                                          int iters, int form) {
    __shared__ unsigned s_xcc, s_rank, s_n, s_ok;
    const int nb = gridDim.x;
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        x &= 7u;
        s_xcc = x;
        s_rank = atomicAdd(&ctl->rank_next[x], 1u);
        atomicAdd(&ctl->census[x], 1u);
        slot_of[x * 256 + s_rank] = blockIdx.x;            // rank -> workgroup of this XCC
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicAdd(&ctl->all_in, 1u);
        s_ok = spin_until(&ctl->all_in, (unsigned)nb, &ctl->timeout);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_n = ld_relaxed(&ctl->census[x]);
    }
    __syncthreads();
    if (!s_ok) return;
    const unsigned xcc = s_xcc, rank = s_rank, n = form == 3 ? (unsigned)nb : s_n;
    unsigned* cnt = form == 3 ? ctl->cnt_all : ctl->cnt[xcc];
    // the workgroup whose record I check: the next rank of my XCC (form 3: the next workgroup id)
    const unsigned peer = form == 3 ? (blockIdx.x + 1) % nb : ld_relaxed(&slot_of[xcc * 256 + (rank + 1) % s_n]);
    unsigned* mine = rec + (size_t)blockIdx.x * 32;
    const unsigned* theirs = rec + (size_t)peer * 32;
    unsigned stale = 0;
    for (int r = 1; r <= iters; ++r) {
        // publish
        if (form == 0) {
            if (threadIdx.x < 32) mine[threadIdx.x] = (unsigned)r;
            __syncthreads();
            if (threadIdx.x == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if (form == 1 || form == 3) {
            if (threadIdx.x < 32) __hip_atomic_store(mine + threadIdx.x, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // barrier among the n workgroups that share `cnt`
        if (threadIdx.x == 0) {
            atomicAdd(cnt, 1u);
            s_ok = spin_until(cnt, (unsigned)r * n, &ctl->timeout);
            if (form == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (!s_ok) break;
        // consume
        if (form == 0) {
            if (threadIdx.x < 32 && theirs[threadIdx.x] != (unsigned)r && theirs[threadIdx.x] != (unsigned)r + 1) ++stale;
        } else if (form == 1 || form == 3) {
            if (threadIdx.x < 32) {
                const unsigned v = ld_relaxed(theirs + threadIdx.x);
                if (v != (unsigned)r && v != (unsigned)r + 1) ++stale;   // (the peer may already have published the next round)
            }
        }
    }
    if (stale) atomicAdd(&ctl->stale, stale);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int dev = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    Ctl* ctl;
    unsigned *rec, *slot_of;
    CHECK(hipMalloc(&ctl, sizeof(Ctl)));
    CHECK(hipMalloc(&rec, (size_t)cus * 32 * 4));
    CHECK(hipMalloc(&slot_of, 8 * 256 * 4));
    const char* names[4] = {"per-XCC barrier, plain stores + release / acquire fences", "per-XCC barrier, sc1 stores / sc1 loads, no fences",
                            "per-XCC barrier alone (arrive + poll)", "ONE counter for all workgroups, sc1 stores / loads"};
    for (int rep = 0; rep < 2; ++rep)
        for (int form = 0; form < 4; ++form) {
            CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
            CHECK(hipMemset(rec, 0, (size_t)cus * 32 * 4));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(xcc_barrier_kernel, dim3(cus), dim3(256), 0, 0, ctl, rec, slot_of, iters, form);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            Ctl h;
            CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
            printf("xcc_barrier form %d (%s): %d workgroups, per XCC [%u %u %u %u %u %u %u %u], %d rounds: %.3f us per round, "
                   "stale records %u, timeout %u\n", form, names[form], cus, h.census[0], h.census[1], h.census[2], h.census[3],
                   h.census[4], h.census[5], h.census[6], h.census[7], iters, ms * 1e3 / iters, h.stale, h.timeout);
            fflush(stdout);
        }
    return 0;
}
