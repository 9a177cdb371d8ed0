// cu_mask_map.hip — which physical CUs does a hipExtStreamCreateWithCUMask bit select on this box?
// Every workgroup records (XCC_ID, SE, SH, CU) from the hardware-id registers; the host prints, per mask, how many distinct CUs ran
// workgroups and how they spread over the XCDs.  Needed before partitioning the chip between the MFMA-bound and the HBM-bound kernels
// (DESIGN.md §4.5): two masks must be physically disjoint and each must span all eight XCDs (L2 slices / memory channels).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/cu_mask_map tools/ubench/cu_mask_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_kernel(unsigned *out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that the dispatcher has to use every CU the mask allows
    unsigned long long t0 = clock64();
    while (clock64() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

struct Cu { int xcc, se, sh, cu; bool operator<(const Cu &o) const { return xcc != o.xcc ? xcc < o.xcc : se != o.se ? se < o.se : sh != o.sh ? sh < o.sh : cu < o.cu; } };

static std::set<Cu> run(const std::vector<int> &bits, unsigned *d_out, std::vector<unsigned> &h, bool *ok)
{
    unsigned words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b : bits) words[b >> 5] |= 1u << (b & 31);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, words);
    *ok = e == hipSuccess;
    if (!*ok) { printf("   hipExtStreamCreateWithCUMask -> %s\n", hipGetErrorString(e)); return {}; }
    const int nwg = 8192;
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(64), 0, s, d_out, 20000);
    CHECK(hipStreamSynchronize(s));
    CHECK(hipMemcpy(h.data(), d_out, nwg * 2 * sizeof(unsigned), hipMemcpyDeviceToHost));
    CHECK(hipStreamDestroy(s));
    std::set<Cu> cus;
    for (int i = 0; i < nwg; ++i) {
        const unsigned hw = h[2 * i], x = h[2 * i + 1];
        cus.insert(Cu{(int)(x & 15), (int)((hw >> 13) & 7), (int)((hw >> 12) & 1), (int)((hw >> 8) & 15)});
    }
    return cus;
}

static void summary(const char *name, const std::set<Cu> &cus)
{
    std::map<int, int> per;
    for (auto &c : cus) per[c.xcc]++;
    printf("%-28s distinct CUs %3zu | per XCC:", name, cus.size());
    for (auto &p : per) printf(" %d:%d", p.first, p.second);
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    unsigned *d_out;
    CHECK(hipMalloc(&d_out, 8192 * 2 * sizeof(unsigned)));
    std::vector<unsigned> h(8192 * 2);
    bool ok;
    auto range = [](int a, int b, int step = 1) { std::vector<int> v; for (int i = a; i < b; i += step) v.push_back(i); return v; };
    printf("# single bits: (xcc, se, sh, cu) that ran workgroups\n");
    for (int b : {0, 1, 2, 3, 7, 8, 9, 15, 16, 31, 32, 33, 63, 64, 127, 128, 255}) {
        auto cus = run({b}, d_out, h, &ok);
        // an XCC whose slice of the mask is all zero runs UNRESTRICTED: report the XCCs that were restricted to few CUs
        std::map<int, std::vector<Cu>> per;
        for (auto &c : cus) per[c.xcc].push_back(c);
        printf("bit %3d -> %3zu CUs;", b, cus.size());
        for (auto &p : per) {
            if (p.second.size() > 4) continue;
            for (auto &c : p.second) printf(" (xcc %d, se %d, sh %d, cu %d)", c.xcc, c.se, c.sh, c.cu);
        }
        printf("   [other XCCs: all their CUs]\n");
    }
    printf("# ranges and patterns\n");
    for (int n : {32, 64, 96, 104, 112, 128, 192, 256}) {
        char nm[64];
        snprintf(nm, sizeof nm, "bits [0, %d)", n);
        auto lo = run(range(0, n), d_out, h, &ok);
        summary(nm, lo);
        if (n < 256) {
            snprintf(nm, sizeof nm, "bits [%d, 256)", n);
            auto hi = run(range(n, 256), d_out, h, &ok);
            summary(nm, hi);
            int both = 0;
            for (auto &c : lo) both += hi.count(c);
            printf("   -> CUs in both: %d\n", both);
        }
    }
    summary("every 2nd bit", run(range(0, 256, 2), d_out, h, &ok));
    summary("every 4th bit", run(range(0, 256, 4), d_out, h, &ok));
    summary("every 8th bit", run(range(0, 256, 8), d_out, h, &ok));
    summary("bits [0,8) u [64,72)", run([&] { auto v = range(0, 8); auto w = range(64, 72); v.insert(v.end(), w.begin(), w.end()); return v; }(), d_out, h, &ok));
    return 0;
}
