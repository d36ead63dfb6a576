// Scratch microbenchmark (GPU box), round 5: what does an ORDERED WORK QUEUE of (step, tile) items cost on this chip?  The barrier-free
// form of the step (profiles/r05_step_tile.txt) hands the steps of a tile from one persistent wavefront to whichever takes the
// tile's next item: the tile's state goes through memory, a flag says "step s done".  This measures that primitive alone -- every
// item reads and rewrites its tile's TILE_WORDS words (x += 1), runs WORK dependent VALU instructions, publishes -- for the ways of
// publishing, with the tiles pinned to an XCD (a queue per XCD, the wavefront reads its XCC_ID) or not, and checks the result (every
// word must equal the step count: a stale read shows up as a smaller number).
//   mode 0  no dependence at all (item order only): the ceiling
//   mode 1  agent-scope release (buffer_wbl2 sc1) + sc1 flag; waiter: sc1 polls by lane 0, then agent-scope acquire (buffer_inv sc1)
//   mode 2  stores waited for (s_waitcnt vmcnt(0): they are in the XCD's L2) + sc1 flag; waiter as in mode 1 -- sound only within an XCD
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_queue.hip -o tools/ubench_queue.bin && tools/ubench_queue.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define TILE_WORDS 2048 // 8 KB of state per tile (a 64-body tile of the stepper reads and writes about that much per step)
#define NXCD 8
template <int WORK> __device__ __forceinline__ float chain(float a, float b, float c) {
#pragma unroll 16
    for (int i = 0; i < WORK; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    return a;
}
__device__ __forceinline__ unsigned xcc_id() { return (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; } // HW_REG_XCC_ID[3:0]

// q layout: per queue (1 global, or one per XCD): [0] next item, [1] abort, [2] polls (diagnostic); done[] per tile behind all queues
template <int WORK>
__global__ void __launch_bounds__(64, 4) queue_kernel(int *state, int *q, int *done, float *sink, int nTiles, int nSteps, int mode, int pinned,
                                                      unsigned *xcdSeen) {
    const int lane = threadIdx.x;
    const unsigned xcd = xcc_id();
    if (lane == 0) atomicAdd(&xcdSeen[xcd & 15u], 1u);
    int *myq = q + (pinned ? (int)(xcd % NXCD) * 16 : 0);
    // pinned: queue x holds the tiles t with t % NXCD == x
    const int myTiles = pinned ? (nTiles - (int)(xcd % NXCD) + NXCD - 1) / NXCD : nTiles;
    const unsigned total = (unsigned)myTiles * (unsigned)nSteps;
    float a = 1.0f + lane * 1e-7f;
    for (;;) {
        unsigned item = 0;
        if (lane == 0) item = __hip_atomic_load(&myq[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? total : (unsigned)atomicAdd(&myq[0], 1);
        item = (unsigned)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= total) break;
        const int step = (int)(item / (unsigned)myTiles);
        const int local = (int)(item % (unsigned)myTiles);
        const int tile = pinned ? local * NXCD + (int)(xcd % NXCD) : local;
        if (mode != 0 && step > 0) {
            if (lane == 0) {
                unsigned spins = 0;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz
                while (__hip_atomic_load(&done[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < step) {
                    __builtin_amdgcn_s_sleep(4);
                    ++spins;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) { // 0.2 s: a lost hand-over must not hang the GPU
                        __hip_atomic_store(&myq[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
                atomicAdd(&myq[2], (int)spins);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        int *t = state + (size_t)tile * TILE_WORDS;
#pragma unroll 4
        for (int w = lane; w < TILE_WORDS; w += 64) t[w] += 1;
        a = chain<WORK>(a, 0.999f, 1e-9f);
        if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&done[tile], step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sink[blockIdx.x * 64 + lane] = a;
}

template <int WORK> int run(int nTiles, int nSteps, int grid) {
    int *state, *q, *done;
    float *sink;
    unsigned *seen;
    CHECK(hipMalloc(&state, (size_t)nTiles * TILE_WORDS * sizeof(int)));
    CHECK(hipMalloc(&q, NXCD * 16 * sizeof(int)));
    CHECK(hipMalloc(&done, nTiles * sizeof(int)));
    CHECK(hipMalloc(&sink, (size_t)grid * 64 * sizeof(float)));
    CHECK(hipMalloc(&seen, 16 * sizeof(unsigned)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<int> host((size_t)nTiles * TILE_WORDS);
    for (int pinned : {0, 1})
        for (int mode : {0, 1, 2}) {
            if (mode == 0 && pinned) continue;
            float best = 1e30f;
            long bad = 0, polls = 0;
            int aborted = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemset(state, 0, (size_t)nTiles * TILE_WORDS * sizeof(int)));
                CHECK(hipMemset(q, 0, NXCD * 16 * sizeof(int)));
                CHECK(hipMemset(done, 0, nTiles * sizeof(int)));
                CHECK(hipMemset(seen, 0, 16 * sizeof(unsigned)));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(queue_kernel<WORK>, dim3(grid), dim3(64), 0, 0, state, q, done, sink, nTiles, nSteps, mode, pinned, seen);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                if (mode != 0) { // (mode 0 has no ordering between the steps of a tile: its sums are not meaningful)
                    CHECK(hipMemcpy(host.data(), state, host.size() * sizeof(int), hipMemcpyDeviceToHost));
                    bad = 0;
                    for (size_t i = 0; i < host.size(); ++i) bad += host[i] != nSteps;
                }
                int hq[NXCD * 16];
                CHECK(hipMemcpy(hq, q, sizeof(hq), hipMemcpyDeviceToHost));
                polls = 0;
                aborted = 0;
                for (int x = 0; x < NXCD; ++x) { polls += hq[x * 16 + 2]; aborted |= hq[x * 16 + 1]; }
            }
            unsigned hs[16];
            CHECK(hipMemcpy(hs, seen, sizeof(hs), hipMemcpyDeviceToHost));
            int xcds = 0;
            for (int x = 0; x < 16; ++x) xcds += hs[x] != 0;
            const double items = (double)nTiles * nSteps;
            printf("work %4d  tiles %5d x %3d steps  grid %4d  %s mode %d: %8.3f ms  %6.3f us per item-slot  polls/item %.2f  wrong words %ld%s  (XCC ids seen: %d)\n",
                   WORK, nTiles, nSteps, grid, pinned ? "per-XCD queues" : "one queue     ", mode, best, best * 1e3 / items * grid, (double)polls / items,
                   bad, aborted ? "  ABORTED (a wait ran into its bound)" : "", xcds);
            fflush(stdout);
        }
    (void)hipFree(state); (void)hipFree(q); (void)hipFree(done); (void)hipFree(sink); (void)hipFree(seen);
    return 0;
}

int main() {
    // 7 788 tiles like config 3's population; 20 steps per call; ~1 200 and ~6 000 dependent instructions per item (the stepper's
    // tile-steps are ~100 000: the hand-over cost is what is measured here, so the work is kept small against it)
    for (int grid : {1024, 4096}) {
        if (run<1200>(7788, 20, grid)) return 1;
        if (run<6000>(7788, 20, grid)) return 1;
    }
    return 0;
}
