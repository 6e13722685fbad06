// tests/cpp/myicp_surface.cpp -- exercises the C++ surface BASELINE.json's north_star names on the MyICP class
// (include/myicp.h, replacing ICP/myicp.h:14-19): setInputSource / setInputTarget with packed arrays, align(out, guess),
// getFinalTransformation, GetAlignedSrcCloud -- twice on the same object, in PAPER + TREE and in QUIRKS + IDENTITY.
//
//   myicp_surface <dir>
// reads   <dir>/src.f32 src_n.f32 tgt.f32 tgt_n.f32     packed float32 [n][3] (written by tests/test_gpu_parity.py)
//         <dir>/guess.f32                               16 floats, row-major 4x4
// writes  <dir>/out_paper_tree.f32, out_quirks_identity.f32        the 4x4 `align` returned
//         <dir>/aligned_paper_tree.f32                              GetAlignedSrcCloud() as packed xyz
// and checks by itself (exit code != 0 on failure): align's out == getFinalTransformation() == lastResult().transform,
// GetAlignedSrcCloud() == X * source recomputed here, a second align on the same object works, getters hand out the clouds
// that were set.  The numbers are compared with the oracle's golden values by the Python test.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "myicp.h"

static std::vector<float> slurp(const std::string &path)
{
    std::vector<float> v;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    v.resize((size_t)bytes / sizeof(float));
    if (std::fread(v.data(), sizeof(float), v.size(), f) != v.size()) { std::fprintf(stderr, "short read on %s\n", path.c_str()); std::exit(2); }
    std::fclose(f);
    return v;
}

static void dump(const std::string &path, const float *p, size_t n)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f || std::fwrite(p, sizeof(float), n, f) != n) { std::fprintf(stderr, "cannot write %s\n", path.c_str()); std::exit(2); }
    std::fclose(f);
}

#define CHECK(cond)                                                                     \
    do {                                                                                \
        if (!(cond)) { std::fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 2) { std::fprintf(stderr, "usage: %s <dir>\n", argv[0]); return 64; }
    const std::string dir = std::string(argv[1]) + "/";
    const std::vector<float> src = slurp(dir + "src.f32"), src_n = slurp(dir + "src_n.f32"), tgt = slurp(dir + "tgt.f32"), tgt_n = slurp(dir + "tgt_n.f32"),
                             guess = slurp(dir + "guess.f32");
    CHECK(src.size() % 3 == 0 && src.size() == src_n.size() && tgt.size() == tgt_n.size() && guess.size() == 16);
    const size_t ns = src.size() / 3, nt = tgt.size() / 3;

    MyICP icp;
    icp.setVerbose(false);
    icp.setInputSource(src.data(), src_n.data(), ns);
    icp.setInputTarget(tgt.data(), tgt_n.data(), nt);
    CHECK(icp.GetSrcCloud()->points.size() == ns && icp.GetTgtCloud()->points.size() == nt);
    CHECK(icp.GetSrcCloud()->points[ns - 1].z == src[3 * (ns - 1) + 2] && icp.GetTgtCloud()->points[0].x == tgt[0]);

    // ---- paper-correct arithmetic, exact nearest neighbours, starting from a guess
    icp.setMode(SYMMICP_MODE_PAPER);
    icp.setCorrespondence(SYMMICP_CORR_TREE);
    icp.setMaximumIterations(30);
    float out[16];
    int st = icp.align(out, guess.data());
    CHECK(st == SYMMICP_OK);
    CHECK(icp.lastResult().status == SYMMICP_OK && icp.lastResult().iters > 0);
    CHECK(std::memcmp(out, icp.getFinalTransformation(), sizeof(out)) == 0);
    CHECK(std::memcmp(out, icp.lastResult().transform, sizeof(out)) == 0);
    dump(dir + "out_paper_tree.f32", out, 16);
    {
        pcl::PointCloud<PointT>::Ptr moved = icp.GetAlignedSrcCloud();
        CHECK(moved && moved->points.size() == ns);
        std::vector<float> xyz(3 * ns);
        for (size_t i = 0; i < ns; i++) {
            const float x = src[3 * i], y = src[3 * i + 1], z = src[3 * i + 2];
            const float ex = ((out[0] * x + out[1] * y) + out[2] * z) + out[3];
            const float ey = ((out[4] * x + out[5] * y) + out[6] * z) + out[7];
            const float ez = ((out[8] * x + out[9] * y) + out[10] * z) + out[11];
            const PointT &p = moved->points[i];
            CHECK(std::fabs(p.x - ex) <= 1e-5f * (1.f + std::fabs(ex)) && std::fabs(p.y - ey) <= 1e-5f * (1.f + std::fabs(ey)) && std::fabs(p.z - ez) <= 1e-5f * (1.f + std::fabs(ez)));
            xyz[3 * i] = p.x; xyz[3 * i + 1] = p.y; xyz[3 * i + 2] = p.z;
        }
        dump(dir + "aligned_paper_tree.f32", xyz.data(), xyz.size());
        // the source cloud itself is not touched (the reference never writes its result back, myicp.cpp:109-111)
        CHECK(icp.GetSrcCloud()->points[0].x == src[0] && icp.GetSrcCloud()->points[ns - 1].y == src[3 * (ns - 1) + 1]);
    }

    // ---- the reference as written, on the same object (needs N_s == N_t: func.cpp:21)
    icp.setMode(SYMMICP_MODE_QUIRKS);
    icp.setCorrespondence(SYMMICP_CORR_IDENTITY);
    icp.setMaximumIterations(10);
    icp.setDiffThreshold(1.0f);
    float out2[16];
    st = icp.align(out2, nullptr);
    if (ns == nt) {
        CHECK(st == SYMMICP_OK);
        CHECK(icp.lastResult().iters == 10);
        CHECK(std::memcmp(out2, icp.getFinalTransformation(), sizeof(out2)) == 0);
        dump(dir + "out_quirks_identity.f32", out2, 16);
    } else {
        CHECK(st == SYMMICP_ERR_SIZE);
    }
    // a cloud edited through the getter is what the next run aligns (the reference recomputes everything from cloud_src)
    const float keep = icp.GetSrcCloud()->points[0].x;
    icp.GetSrcCloud()->points[0].x = keep + 1.0f;
    float out3[16];
    st = icp.align(out3, nullptr);
    CHECK(ns != nt || st == SYMMICP_OK);
    CHECK(ns != nt || std::memcmp(out2, out3, sizeof(out2)) != 0);
    std::printf("myicp_surface: ok (%zu source, %zu target points; paper/tree %d iterations)\n", ns, nt, 0);
    return 0;
}
