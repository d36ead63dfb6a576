// Scratch microbenchmark (GPU box), round 5: what does a hand-off between TWO wavefronts of one workgroup cost?  The one
// design idea left against the chain wall of DESIGN.md 5 is a role split (a J-wavefront for a tile's joint slots, a C-wavefront
// for its contact sub-slots, body velocities and per-body sequence counters in LDS): every slot would start with a wait for the
// other wavefront's counter and end with publishing its own.  This measures that primitive alone -- a ping-pong through one LDS
// word between the two wavefronts of a block, `work` dependent VALU instructions per turn -- against the same instruction count
// run by one wavefront, for 1 and 4 blocks per CU-quarter (the velocity kernel's occupancy).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_handoff.hip -o /tmp/ubench_handoff && /tmp/ubench_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int WORK> __device__ __forceinline__ float chain(float a, float b, float c) {
#pragma unroll
    for (int i = 0; i < WORK; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    return a;
}
// two wavefronts, strictly alternating turns: wave w runs its `WORK` instructions when turn % 2 == w, then publishes turn + 1
template <int WORK> __global__ void __launch_bounds__(128) pingpong(float *out, unsigned long long *stamps, int turns) {
    __shared__ volatile int turn;
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) turn = 0;
    __syncthreads();
    float a = 1.0f + threadIdx.x * 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = wave; t < turns; t += 2) {
        while (turn != t) __builtin_amdgcn_s_sleep(1);
        a = chain<WORK>(a, 0.999f, 1e-9f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((threadIdx.x & 63) == 0) turn = t + 1;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 128 + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) stamps[2 * blockIdx.x + wave] = t1 - t0;
}
// two wavefronts that BOTH run `WORK` instructions per turn and meet at a workgroup barrier after each: what s_barrier itself costs
template <int WORK> __global__ void __launch_bounds__(128) barrier_pair(float *out, unsigned long long *stamps, int turns) {
    float a = 1.0f + threadIdx.x * 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < turns; ++t) {
        a = chain<WORK>(a, 0.999f, 1e-9f);
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 128 + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) stamps[2 * blockIdx.x + (threadIdx.x >> 6)] = t1 - t0;
}
// the same number of turns by ONE wavefront
template <int WORK> __global__ void __launch_bounds__(64) serial(float *out, unsigned long long *stamps, int turns) {
    float a = 1.0f + threadIdx.x * 1e-7f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < turns; ++t) a = chain<WORK>(a, 0.999f, 1e-9f);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a;
    if (threadIdx.x == 0) stamps[blockIdx.x] = t1 - t0;
}
template <int WORK> int run(float *out, unsigned long long *st, int ncu) {
    const int turns = 2000;
    for (int per_cu : {1, 8}) {
        const int blocks = ncu * per_cu;
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(pingpong<WORK>, dim3(blocks), dim3(128), 0, 0, out, st, turns);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(2 * blocks);
        CHECK(hipMemcpy(h.data(), st, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> pp;
        for (int b = 0; b < blocks; ++b) pp.push_back((double)std::max(h[2 * b], h[2 * b + 1]) / turns);
        std::sort(pp.begin(), pp.end());
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(serial<WORK>, dim3(blocks * 2), dim3(64), 0, 0, out, st, turns);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> g(2 * blocks);
        CHECK(hipMemcpy(g.data(), st, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> se;
        for (int b = 0; b < 2 * blocks; ++b) se.push_back((double)g[b] / turns);
        std::sort(se.begin(), se.end());
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(barrier_pair<WORK>, dim3(blocks), dim3(128), 0, 0, out, st, turns);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), st, 2 * blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<double> bp;
        for (int b = 0; b < blocks; ++b) bp.push_back((double)std::max(h[2 * b], h[2 * b + 1]) / turns);
        std::sort(bp.begin(), bp.end());
        printf("%3d dependent v_fma per turn, %d block(s) per CU: one wavefront %.0f ticks per turn; two wavefronts alternating through LDS %.0f "
               "(median; p90 %.0f) -> the hand-off costs %.0f ticks; two wavefronts side by side + s_barrier per turn %.0f -> the barrier costs %.0f\n",
               WORK, per_cu, se[se.size() / 2], pp[pp.size() / 2], pp[pp.size() * 9 / 10], pp[pp.size() / 2] - se[se.size() / 2],
               bp[bp.size() / 2], bp[bp.size() / 2] - se[se.size() / 2]);
    }
    return 0;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("%s: %d CUs\n", prop.name, ncu);
    float *out; unsigned long long *st;
    CHECK(hipMalloc(&out, (size_t)ncu * 16 * 128 * sizeof(float))); CHECK(hipMalloc(&st, (size_t)ncu * 16 * 2 * sizeof(unsigned long long)));
    if (run<32>(out, st, ncu)) return 1;
    if (run<100>(out, st, ncu)) return 1;
    return 0;
}
