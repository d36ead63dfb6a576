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
// mode 4 (round 6): EIGHT independent v_fma_f32 accumulators, round-robin -- no instruction depends on any of the seven before it,
//         so the figure is the chip's VALU issue ceiling itself (MI355X_MICROARCH.md: wave64 v_fma_f32 2 cycles on a SIMD-32, one
//         wave alone 4), not the ceiling of dependent chains that the rows above measure
template <int MODE>
__global__ void __launch_bounds__(64) chain_kernel(float *out, long long *cycles, int trips, int lanes) {
    __shared__ float lds[64 * 4];
    const int lane = threadIdx.x;
    float a = 1.0f + lane * 1e-7f, b = 0.999f, c = 1e-9f, a2 = 1.1f;
    v2f p = {a, a2}, q = {b, b}, r = {c, c};
    float e0 = a, e1 = a2, e2 = a + 1.0f, e3 = a + 2.0f, e4 = a + 3.0f, e5 = a + 4.0f, e6 = a + 5.0f, e7 = a + 6.0f;
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
            } else if (MODE == 4) {
#define FMA8 asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                          "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" \
                          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7) : "v"(b), "v"(c));
                REP8(FMA8 FMA8)   // 16 x 8 = 128 instructions per trip
#undef FMA8
            } else {
                REP8(REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a) : "v"(b), "v"(c));)
                     lds[lane] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                     a = lds[lane ^ 1];)
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    out[blockIdx.x * 64 + lane] = a + a2 + p.x + p.y + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
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
    if (run<4>("8 independent v_fma_f32 streams", 64, out, cyc, nsimd)) return 1;
    if (run<4>("8 independent v_fma_f32 streams", 12, out, cyc, nsimd)) return 1;
    return 0;
}
