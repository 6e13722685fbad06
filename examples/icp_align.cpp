// examples/icp_align.cpp -- command-line registration of two PCD files on one MI355X through the MyICP class.
//
//   icp_align [options] [source.pcd target.pcd]
//     --mode quirks|paper      arithmetic: the reference as written (default) or the paper-correct form
//     --corr identity|tree     pairing: by row (default, what the reference does) or exact nearest neighbours
//     --iters N                iteration cap            (default 10, ICP/myicp.cpp:6)
//     --threshold D            stop once the summed pair distance is <= D   (default 1.0, ICP/myicp.cpp:6)
//     --out aligned.pcd        write the source moved by the result (the reference only prints its result)
//     --quiet                  no per-iteration lines
//   Without file names it registers cat.pcd to cat_out.pcd from the working directory: the reference's own run.
// Exit code: the symmicp status of the alignment (0 = ok).
//
// The drop-in property itself -- the reference's ICP/main.cpp compiling byte-unchanged against include/myicp.h -- is
// checked in the build container by tests/test_abi.py; this program is the repo's own driver for the same class.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "myicp.h"

static int usage(const char *argv0, const char *complaint)
{
    std::fprintf(stderr, "%s\nusage: %s [--mode quirks|paper] [--corr identity|tree] [--iters N] [--threshold D] [--out file.pcd] [--quiet] [source.pcd target.pcd]\n",
                 complaint, argv0);
    return 64;
}

int main(int argc, char **argv)
{
    std::vector<std::string> files;
    std::string out_path;
    MyICP icp;
    for (int k = 1; k < argc; k++) {
        const std::string a = argv[k];
        auto value = [&](const char *what) -> const char * {
            if (k + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", what); std::exit(64); }
            return argv[++k];
        };
        if (a == "--mode") {
            const std::string v = value("--mode");
            if (v == "quirks") icp.setMode(SYMMICP_MODE_QUIRKS);
            else if (v == "paper") icp.setMode(SYMMICP_MODE_PAPER);
            else return usage(argv[0], "unknown --mode");
        } else if (a == "--corr") {
            const std::string v = value("--corr");
            if (v == "identity") icp.setCorrespondence(SYMMICP_CORR_IDENTITY);
            else if (v == "tree") icp.setCorrespondence(SYMMICP_CORR_TREE);
            else return usage(argv[0], "unknown --corr");
        } else if (a == "--iters") icp.setMaximumIterations(std::atoi(value("--iters")));
        else if (a == "--threshold") icp.setDiffThreshold((float)std::atof(value("--threshold")));
        else if (a == "--out") out_path = value("--out");
        else if (a == "--quiet") icp.setVerbose(false);
        else if (!a.empty() && a[0] == '-') return usage(argv[0], ("unknown option " + a).c_str());
        else files.push_back(a);
    }
    if (files.empty()) files = {"cat.pcd", "cat_out.pcd"};
    if (files.size() != 2) return usage(argv[0], "expected two PCD files");

    icp.LoadCloud(files[0], files[1]);
    if (icp.GetSrcCloud()->points.empty() || icp.GetTgtCloud()->points.empty()) {
        std::fprintf(stderr, "%s\n", icp.lastError()[0] ? icp.lastError() : "empty cloud");
        return SYMMICP_ERR_IO;
    }
    icp.RegisterSymm();
    const symmicp_result &r = icp.lastResult();
    if (r.status != SYMMICP_OK) return r.status;

    if (!out_path.empty()) {
        pcl::PointCloud<PointT>::Ptr moved = icp.GetAlignedSrcCloud();
        std::vector<float> xyz(3 * moved->points.size());
        for (size_t i = 0; i < moved->points.size(); i++) { xyz[3 * i] = moved->points[i].x; xyz[3 * i + 1] = moved->points[i].y; xyz[3 * i + 2] = moved->points[i].z; }
        if (symmicp_pcd_write(out_path.c_str(), xyz.data(), nullptr, moved->points.size(), 0) != 0) {
            std::fprintf(stderr, "cannot write %s\n", out_path.c_str());
            return SYMMICP_ERR_IO;
        }
    }
    return 0;
}
