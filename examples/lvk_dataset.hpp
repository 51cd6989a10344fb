// lvk_dataset.hpp — readers for the ASL/EuRoC directory layout the reference's drivers consume: imu0/data.csv (ns, wx, wy, wz,
// ax, ay, az) and cam0/data.csv (ns, file name).  They do the job of loadImuFile / loadImageList
// (/root/reference/include/utils/DataReader.hpp:31-119) with the same conventions — the first line is a header, time is
// 1e-9 * the integer nanosecond stamp — and two deliberate differences: blank lines (the file's trailing newline) are skipped
// instead of producing a record with stamp 0, and the image name is stripped of any trailing CR / blanks rather than of exactly
// one character.  Host-only.
#ifndef LVK_DATASET_HPP
#define LVK_DATASET_HPP
#include "lvk_larvio.hpp"
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace lvk {

struct ImgInfo { double timeStampToSec; std::string imgName; };

namespace dataset_detail {
inline bool read_lines(const std::string& path, std::vector<std::string>* lines)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::string text; char buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos); if (eol == std::string::npos) eol = text.size();
        size_t end = eol; while (end > pos && (text[end - 1] == '\r' || text[end - 1] == ' ' || text[end - 1] == '\t')) --end;
        lines->push_back(text.substr(pos, end - pos));
        pos = eol + 1;
    }
    return true;
}
}  // namespace dataset_detail

inline bool loadImuFile(const std::string& path, std::vector<ImuData>& out)
{
    std::vector<std::string> lines;
    if (!dataset_detail::read_lines(path, &lines)) return false;
    for (size_t i = 1; i < lines.size(); ++i) {                       // line 0: the column header
        const std::string& L = lines[i]; if (L.empty()) continue;
        ImuData d; const char* p = L.c_str(); char* q = nullptr;
        d.timeStampToSec = 1e-9 * (double)std::strtoll(p, &q, 10);
        double v[6] = {0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 6 && *q == ','; ++k) { p = q + 1; v[k] = std::strtod(p, &q); }
        for (int k = 0; k < 3; ++k) { d.angular_velocity[k] = v[k]; d.linear_acceleration[k] = v[3 + k]; }
        out.push_back(d);
    }
    return true;
}

inline bool loadImageList(const std::string& path, std::vector<ImgInfo>& out)
{
    std::vector<std::string> lines;
    if (!dataset_detail::read_lines(path, &lines)) return false;
    for (size_t i = 1; i < lines.size(); ++i) {
        const std::string& L = lines[i]; if (L.empty()) continue;
        const size_t c = L.find(',');
        ImgInfo info; info.timeStampToSec = 1e-9 * (double)std::strtoll(L.c_str(), nullptr, 10);
        if (c != std::string::npos) { const size_t c2 = L.find(',', c + 1); info.imgName = L.substr(c + 1, c2 == std::string::npos ? std::string::npos : c2 - c - 1); }
        out.push_back(info);
    }
    return true;
}

}  // namespace lvk
#endif
