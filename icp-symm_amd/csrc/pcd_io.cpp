// pcd_io.cpp -- PCD v0.7 reader/writer, the on-disk format either side of the path.
// Replaces the pcl::PCDReader + fromPCLPointCloud2 pair used by MyICP::LoadCloud
// (reference ICP/myicp.cpp:20-31): fields are located BY NAME, x/y/z are kept, other
// fields (label, curvature, ...) are skipped, normal_x/y/z are returned when present.
// Handles the two headers the reference ships (cat.pcd: 3 fields; cat_out.pcd: 8 fields with
// a TYPE U label column) in DATA ascii and DATA binary.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>
#include "symmicp.h"

namespace {
struct Field {
    std::string name;
    int size = 4, count = 1;
    char type = 'F';
    int column = 0;      // first ascii column
    int offset = 0;      // byte offset in a binary record
};

struct Header {
    std::vector<Field> fields;
    long width = -1, height = 1, points = -1;
    bool binary = false;
    int columns = 0, record = 0;
    int find(const char *nm) const
    {
        for (size_t k = 0; k < fields.size(); ++k)
            if (fields[k].name == nm) return (int)k;
        return -1;
    }
};

bool parse_header(FILE *f, Header &h)
{
    char line[8192];
    bool data_seen = false;
    while (std::fgets(line, sizeof line, f)) {
        if (line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        if (!(ss >> key)) continue;
        if (key == "FIELDS") {
            std::string nm;
            while (ss >> nm) { Field fd; fd.name = nm; h.fields.push_back(fd); }
        } else if (key == "SIZE") {
            for (auto &fd : h.fields) if (!(ss >> fd.size)) return false;
        } else if (key == "TYPE") {
            for (auto &fd : h.fields) { std::string t; if (!(ss >> t)) return false; fd.type = t[0]; }
        } else if (key == "COUNT") {
            for (auto &fd : h.fields) if (!(ss >> fd.count)) return false;
        } else if (key == "WIDTH") ss >> h.width;
        else if (key == "HEIGHT") ss >> h.height;
        else if (key == "POINTS") ss >> h.points;
        else if (key == "DATA") {
            std::string kind;
            ss >> kind;
            if (kind == "ascii") h.binary = false;
            else if (kind == "binary") h.binary = true;
            else return false;     // binary_compressed is not produced by anything on this path
            data_seen = true;
            break;
        }
    }
    if (!data_seen || h.fields.empty() || h.fields.size() > 256) return false;
    // the header is untrusted input: every SIZE / COUNT / TYPE has to be one this reader knows how to place and load, the
    // record has to stay small, and POINTS has to agree with WIDTH x HEIGHT when both are given
    if (h.width >= 0 && (h.height < 0 || h.width > (long)0x7fffffff || h.height > (long)0x7fffffff)) return false;
    const long wh = (h.width >= 0) ? h.width * h.height : -1;
    if (h.points < 0) h.points = wh;
    else if (wh >= 0 && wh != h.points) return false;
    if (h.points < 0 || h.points > (long)0x7fffffff) return false;
    for (auto &fd : h.fields) {
        if (fd.count < 1 || fd.count > 4096) return false;
        if (fd.size != 1 && fd.size != 2 && fd.size != 4 && fd.size != 8) return false;
        if (fd.type != 'F' && fd.type != 'U' && fd.type != 'I') return false;
        if (fd.type == 'F' && fd.size < 4) return false;
        fd.column = h.columns; fd.offset = h.record;
        h.columns += fd.count; h.record += fd.count * fd.size;
        if (h.record > (1 << 20) || h.columns > (1 << 16)) return false;
    }
    return h.record > 0;
}

// a field this reader can turn into a float (the x / y / z / normal_* fields have to be)
bool loadable(const Field &fd)
{
    if (fd.type == 'F') return fd.size == 4 || fd.size == 8;
    return fd.size == 1 || fd.size == 2 || fd.size == 4;
}

float load_scalar(const unsigned char *p, const Field &fd)
{
    if (fd.type == 'F' && fd.size == 4) { float v; std::memcpy(&v, p, 4); return v; }
    if (fd.type == 'F' && fd.size == 8) { double v; std::memcpy(&v, p, 8); return (float)v; }
    if (fd.type == 'U' && fd.size == 4) { uint32_t v; std::memcpy(&v, p, 4); return (float)v; }
    if (fd.type == 'I' && fd.size == 4) { int32_t v; std::memcpy(&v, p, 4); return (float)v; }
    if (fd.type == 'U' && fd.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return (float)v; }
    if (fd.type == 'I' && fd.size == 2) { int16_t v; std::memcpy(&v, p, 2); return (float)v; }
    if (fd.type == 'U' && fd.size == 1) return (float)*p;
    if (fd.type == 'I' && fd.size == 1) return (float)*(const signed char *)p;
    return 0.f;
}
}  // namespace

extern "C" long symmicp_pcd_read(const char *path, float *xyz, float *nrm, size_t cap, int *has_normals)
{
    if (!path) return -SYMMICP_ERR_ARG;
    FILE *f = std::fopen(path, "rb");
    if (!f) return -SYMMICP_ERR_IO;
    Header h;
    if (!parse_header(f, h)) { std::fclose(f); return -SYMMICP_ERR_IO; }
    const int ix = h.find("x"), iy = h.find("y"), iz = h.find("z");
    const int inx = h.find("normal_x"), iny = h.find("normal_y"), inz = h.find("normal_z");
    if (ix < 0 || iy < 0 || iz < 0) { std::fclose(f); return -SYMMICP_ERR_IO; }
    const bool hn = inx >= 0 && iny >= 0 && inz >= 0;
    {
        bool ok = loadable(h.fields[ix]) && loadable(h.fields[iy]) && loadable(h.fields[iz]);
        if (hn) ok = ok && loadable(h.fields[inx]) && loadable(h.fields[iny]) && loadable(h.fields[inz]);
        if (!ok) { std::fclose(f); return -SYMMICP_ERR_IO; }
    }
    if (has_normals) *has_normals = hn ? 1 : 0;
    if (!xyz) { std::fclose(f); return h.points; }
    if ((size_t)h.points > cap) { std::fclose(f); return -SYMMICP_ERR_SIZE; }
    const Field &fx = h.fields[ix], &fy = h.fields[iy], &fz = h.fields[iz];
    if (h.binary) {
        std::vector<unsigned char> rec((size_t)h.record);
        for (long i = 0; i < h.points; ++i) {
            if (std::fread(rec.data(), 1, rec.size(), f) != rec.size()) { std::fclose(f); return -SYMMICP_ERR_IO; }
            xyz[3 * i] = load_scalar(rec.data() + fx.offset, fx);
            xyz[3 * i + 1] = load_scalar(rec.data() + fy.offset, fy);
            xyz[3 * i + 2] = load_scalar(rec.data() + fz.offset, fz);
            if (nrm && hn) {
                nrm[3 * i] = load_scalar(rec.data() + h.fields[inx].offset, h.fields[inx]);
                nrm[3 * i + 1] = load_scalar(rec.data() + h.fields[iny].offset, h.fields[iny]);
                nrm[3 * i + 2] = load_scalar(rec.data() + h.fields[inz].offset, h.fields[inz]);
            }
        }
    } else {
        char line[8192];
        std::vector<double> v((size_t)h.columns);
        for (long i = 0; i < h.points; ++i) {
            if (!std::fgets(line, sizeof line, f)) { std::fclose(f); return -SYMMICP_ERR_IO; }
            char *s = line;
            int got = 0;
            while (got < h.columns) {
                char *e;
                double d = std::strtod(s, &e);
                if (e == s) break;
                v[got++] = d;
                s = e;
            }
            if (got < h.columns) { std::fclose(f); return -SYMMICP_ERR_IO; }
            xyz[3 * i] = (float)v[fx.column]; xyz[3 * i + 1] = (float)v[fy.column]; xyz[3 * i + 2] = (float)v[fz.column];
            if (nrm && hn) {
                nrm[3 * i] = (float)v[h.fields[inx].column];
                nrm[3 * i + 1] = (float)v[h.fields[iny].column];
                nrm[3 * i + 2] = (float)v[h.fields[inz].column];
            }
        }
    }
    std::fclose(f);
    return h.points;
}

// writer: the call the reference keeps commented out at main.cpp:51-52 / test.cpp:58 (savePCDFile)
extern "C" int symmicp_pcd_write(const char *path, const float *xyz, const float *nrm, size_t n, int binary)
{
    if (!path || !xyz) return SYMMICP_ERR_ARG;
    FILE *f = std::fopen(path, "wb");
    if (!f) return SYMMICP_ERR_IO;
    std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n");
    if (nrm) std::fprintf(f, "FIELDS x y z normal_x normal_y normal_z\nSIZE 4 4 4 4 4 4\nTYPE F F F F F F\nCOUNT 1 1 1 1 1 1\n");
    else std::fprintf(f, "FIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n");
    std::fprintf(f, "WIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA %s\n", n, n, binary ? "binary" : "ascii");
    for (size_t i = 0; i < n; ++i) {
        if (binary) {
            std::fwrite(xyz + 3 * i, sizeof(float), 3, f);
            if (nrm) std::fwrite(nrm + 3 * i, sizeof(float), 3, f);
        } else if (nrm) {
            std::fprintf(f, "%.9g %.9g %.9g %.9g %.9g %.9g\n", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]);
        } else {
            std::fprintf(f, "%.9g %.9g %.9g\n", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        }
    }
    const bool ok = std::ferror(f) == 0;
    std::fclose(f);
    return ok ? SYMMICP_OK : SYMMICP_ERR_IO;
}
