// Where do the waves of co-resident workgroups land?  Each wave records (XCC, SE, CU, SIMD) and its start / end
// clock; the host prints, per CU, the workgroups that overlapped in time and the SIMD of each of their waves.
//   hipcc --offload-arch=gfx950 -O2 -o placement placement.hip && ./placement [blocks] [threads] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>

struct Rec { unsigned hw, xcc; unsigned long long t0, t1; };

__global__ void probe(Rec* out, int spin) {
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_readcyclecounter();
    double x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
    if (x == 12345.678) lds[threadIdx.x] = 1;
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        Rec r;
        r.hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
        r.xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));
        r.t0 = t0; r.t1 = t1;
        out[blockIdx.x * (blockDim.x >> 6) + wave] = r;
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, threads = argc > 2 ? atoi(argv[2]) : 256;
    const int lds = argc > 3 ? atoi(argv[3]) : 40000, spin = argc > 4 ? atoi(argv[4]) : 20000;
    const int wpb = threads / 64;
    Rec* d; hipMalloc(&d, sizeof(Rec) * blocks * wpb);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, spin);
        hipDeviceSynchronize();
    }
    std::vector<Rec> h(blocks * wpb);
    hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
    // key: (xcc, se, cu) -> list of blocks
    std::map<unsigned, std::vector<int>> cu_blocks;
    for (int b = 0; b < blocks; ++b) {
        const Rec& r = h[b * wpb];
        const unsigned cu = (r.hw >> 8) & 15, sh = (r.hw >> 12) & 1, se = (r.hw >> 13) & 7;
        cu_blocks[(r.xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
    }
    printf("blocks %d x %d threads, %d B LDS: %zu distinct (xcc, se, sh, cu)\n", blocks, threads, lds, cu_blocks.size());
    int shown = 0;
    std::map<std::vector<int>, int> pattern_count;     // SIMD patterns of co-resident blocks
    for (auto& kv : cu_blocks) {
        std::vector<int> pat;
        for (int b : kv.second)
            for (int w = 0; w < wpb; ++w) pat.push_back((h[b * wpb + w].hw >> 4) & 3);
        ++pattern_count[pat];
        if (shown++ < 12) {
            printf("xcc %u se %u sh %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 255, (kv.first >> 4) & 15, kv.first & 15);
            for (int b : kv.second) {
                printf("  wg %4d simd", b);
                for (int w = 0; w < wpb; ++w) printf(" %u", (h[b * wpb + w].hw >> 4) & 3);
                printf(" t0 %llu", (h[b * wpb].t0 / 100) % 1000000);
            }
            printf("\n");
        }
    }
    printf("SIMD patterns (waves of the CU's workgroups in block order) and how many CUs show them:\n");
    std::vector<std::pair<int, std::vector<int>>> pc;
    for (auto& kv : pattern_count) pc.push_back({kv.second, kv.first});
    std::sort(pc.rbegin(), pc.rend());
    for (size_t i = 0; i < pc.size() && i < 12; ++i) {
        printf("  %4d x :", pc[i].first);
        for (int s : pc[i].second) printf(" %d", s);
        printf("\n");
    }
    return 0;
}
