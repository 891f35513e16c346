// stand-in for <pcl/io/pcd_io.h>: a reader for the two plain PCD encodings (DATA ascii / DATA binary, 4-byte fields) that
// tests write with global-lvba_amd/dataset.py; the writers throw.  TEST INFRASTRUCTURE ONLY.
#pragma once
#include "../../lvba_unavailable.h"
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
namespace pcl {
namespace io {
namespace lvba_detail {
inline void put(PointXYZI &p, const std::string &f, float v)
{
    if (f == "x") p.x = v; else if (f == "y") p.y = v; else if (f == "z") p.z = v; else if (f == "intensity") p.intensity = v;
}
} // namespace lvba_detail
inline int loadPCDFile(const std::string &path, PointCloud<PointXYZI> &cloud)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) return -1;
    std::vector<std::string> fields;
    std::vector<int> sizes;
    std::vector<char> types;
    size_t n_points = 0;
    std::string mode, line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream is(line);
        std::string key, tok;
        is >> key;
        if (key == "FIELDS") { while (is >> tok) fields.push_back(tok); }
        else if (key == "SIZE") { int s; while (is >> s) sizes.push_back(s); }
        else if (key == "TYPE") { char c; while (is >> c) types.push_back(c); }
        else if (key == "POINTS") { is >> n_points; }
        else if (key == "DATA") { is >> mode; break; }
    }
    if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size()) return -1;
    cloud.clear();
    cloud.points.reserve(n_points);
    if (mode == "ascii") {
        for (size_t i = 0; i < n_points && std::getline(in, line); ++i) {
            std::istringstream is(line);
            PointXYZI p;
            for (const std::string &f : fields) { float v = 0; is >> v; lvba_detail::put(p, f, v); }
            cloud.points.push_back(p);
        }
    } else if (mode == "binary") {
        size_t stride = 0;
        for (int s : sizes) stride += (size_t)s;
        std::vector<char> rec(stride);
        for (size_t i = 0; i < n_points; ++i) {
            if (!in.read(rec.data(), (std::streamsize)stride)) return -1;
            PointXYZI p;
            size_t off = 0;
            for (size_t k = 0; k < fields.size(); ++k) {
                if (types[k] == 'F' && sizes[k] == 4) { float v; std::memcpy(&v, rec.data() + off, 4); lvba_detail::put(p, fields[k], v); }
                off += (size_t)sizes[k];
            }
            cloud.points.push_back(p);
        }
    } else {
        return -1;
    }
    cloud.width = (uint32_t)cloud.points.size();
    cloud.height = 1;
    return 0;
}
template <class P> int savePCDFileBinary(const std::string &, const PointCloud<P> &) { return 0; } // the export's .pcd copies are dropped
template <class P> int savePCDFileBinaryCompressed(const std::string &, const PointCloud<P> &) { lvba_unavailable("pcl::io::savePCDFileBinaryCompressed"); }
template <class P> int savePCDFile(const std::string &, const PointCloud<P> &) { lvba_unavailable("pcl::io::savePCDFile"); }
} // namespace io
} // namespace pcl
