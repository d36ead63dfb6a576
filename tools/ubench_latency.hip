// Scratch microbenchmark (GPU box): cycles per DEPENDENT VALU instruction of one wavefront as a function of the
// wavefronts resident per SIMD, for v_mul/v_add chains, packed v_pk_mul/v_pk_add chains, chains with an LDS round
// trip, and chains executed with 10 of 64 lanes.  Answers "what does a Gauss-Seidel sweep cost per instruction".
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_latency.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// mode 0: scalar-f32 dependent chain (mul, add alternating)   -> 128 instructions per loop trip
// mode 1: packed dependent chain (v_pk_mul_f32, v_pk_add_f32) -> 128 instructions per loop trip
// mode 2: two independent scalar chains interleaved           -> 128 instructions per loop trip
// mode 3: chain with an LDS write + read every 16 instructions
template <int MODE>
__global__ void __launch_bounds__(64) chain_kernel(float *out, long long *cycles, int trips, int lanes) {
    __shared__ float lds[64 * 4];
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-7f, b = 0.999f, c = 1e-9f, a2 = 1.1f;
    v2f p = {a, a2}, q = {b, b}, r = {c, c};
    long long t0 = 0, t1 = 0;
    if (lane < lanes) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < trips; ++i) {
            if (MODE == 0) {
                REP64(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));)
            } else if (MODE == 1) {
                REP64(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %2" : "+v"(p) : "v"(q), "v"(r));)
            } else if (MODE == 2) {
                REP64(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a), "+v"(a2) : "v"(b));)
            } else {
                REP8(REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));)
                     lds[lane] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                     a = lds[lane ^ 1];)
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * 64 + lane] = a + a2 + p.x + p.y;
    if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char *name, int lanes, float *out, long long *cyc, int nsimd) {
    const int trips = 200;
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        int blocks = nsimd * wps;
        hipLaunchKernelGGL(chain_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, trips, lanes);
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(chain_kernel<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, trips, lanes);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> h(blocks);
        CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
        double mean = 0; long long mx = 0;
        for (auto v : h) { mean += v; mx = v > mx ? v : mx; }
        mean /= blocks;
        const double ninstr = 128.0 * trips;   // + 8 lds pairs in mode 3
        printf("%-34s lanes %2d  waves/SIMD %d: s_memtime ticks per instr %.2f (max %.2f)   kernel %.3f ms -> %.2f ns per instr per wave, %.2f G wave-instr/s chip\n",
               name, lanes, wps, mean / ninstr, mx / ninstr, ms, ms * 1e6 / ninstr, blocks * ninstr / (ms * 1e6));
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int nsimd = prop.multiProcessorCount * 4;
    printf("%s: %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    float *out; long long *cyc;
    CHECK(hipMalloc(&out, nsimd * 8 * 64 * sizeof(float)));
    CHECK(hipMalloc(&cyc, nsimd * 8 * sizeof(long long)));
    if (run<0>("dependent v_mul/v_add", 64, out, cyc, nsimd)) return 1;
    if (run<0>("dependent v_mul/v_add", 10, out, cyc, nsimd)) return 1;
    if (run<1>("dependent v_pk_mul/v_pk_add", 64, out, cyc, nsimd)) return 1;
    if (run<1>("dependent v_pk_mul/v_pk_add", 10, out, cyc, nsimd)) return 1;
    if (run<2>("2 independent v_mul chains", 64, out, cyc, nsimd)) return 1;
    if (run<3>("v_mul/v_add + LDS trip per 16", 64, out, cyc, nsimd)) return 1;
    return 0;
}
