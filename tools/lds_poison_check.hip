// Positive control for the "poison_lds" test hook (csrc/vitlora.hip): LDS keeps its content across kernel launches on gfx950, and a
// fill kernel of 2 x #CUs workgroups that each own a CU's whole 160 KB reaches every CU.  A second kernel of the same shape reads
// the LDS WITHOUT writing it and counts the words that still hold the pattern.
//   hipcc --offload-arch=gfx950 -O2 -o tools/lds_poison_check tools/lds_poison_check.hip && tools/lds_poison_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
constexpr int N = 160 * 1024 / 4;
__global__ __launch_bounds__(256) void fill(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned l[];
    for (int i = threadIdx.x; i < N; i += 256) l[i] = pattern;
    __syncthreads();
    if (sink && l[(threadIdx.x * 97) % N] != pattern) *sink = 1;
}
__global__ __launch_bounds__(256) void probe(unsigned pattern, unsigned* hits, unsigned* hwid) {
    extern __shared__ unsigned l[];
    unsigned n = 0;
    for (int i = threadIdx.x; i < N; i += 256) n += l[i] == pattern;
    atomicAdd(hits + blockIdx.x, n);
    if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); hwid[blockIdx.x] = id; }
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, G = 2 * cus;
    hipFuncSetAttribute((const void*)fill, hipFuncAttributeMaxDynamicSharedMemorySize, N * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, N * 4);
    unsigned *hits, *hw;
    hipMalloc(&hits, G * 4); hipMalloc(&hw, G * 4);
    for (unsigned pat : {0xFFFFFFFFu, 0x12345678u}) {
        hipMemset(hits, 0, G * 4);
        hipLaunchKernelGGL(fill, dim3(G), dim3(256), N * 4, 0, pat, (unsigned*)nullptr);
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), N * 4, 0, pat, hits, hw);
        std::vector<unsigned> h(G), w(G);
        hipMemcpy(h.data(), hits, G * 4, hipMemcpyDeviceToHost); hipMemcpy(w.data(), hw, G * 4, hipMemcpyDeviceToHost);
        long full = 0; std::set<unsigned> where;
        for (int i = 0; i < G; ++i) { full += h[i] == (unsigned)N; where.insert(w[i] & 0xFFFFFF00u); }
        printf("pattern %08x: %ld of %d probe workgroups found all %d LDS words still holding it (distinct HW_ID CU fields seen: %zu, CUs %d)\n",
               pat, full, G, N, where.size(), cus);
    }
    return 0;
}
