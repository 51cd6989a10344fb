// ORACLE / TEST INFRASTRUCTURE ONLY: the handful of OpenCV names the reference's LarVio::loadParameters touches (cv::FileStorage over the
// `%YAML:1.0` files the reference ships, cv::Mat / Rect / Matx33d / Vec3d for the camera-IMU transform, cv2eigen), so that src/larvio.cpp
// can be compiled in place without OpenCV (oracle/Makefile, target `ref`).  The reader understands what those files contain: top-level
// `key: scalar`, one level of nesting (`intrinsics:` + indented keys), and `!!opencv-matrix` nodes with rows / cols / data.
#pragma once
#include <string>
#include <map>
#include <vector>
#include <fstream>
#include <sstream>
#include <cstdlib>
#include <cctype>
#include "../lvref_eigen2.hpp"
namespace cv {
struct Rect { int x, y, width, height; Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {} };
class Mat {
public:
    int rows = 0, cols = 0; std::vector<double> d;
    Mat() {}
    Mat(int r, int c) : rows(r), cols(c), d((size_t)r * c, 0.0) {}
    double at(int i, int j) const { return d[(size_t)i * cols + j]; }
    Mat operator()(const Rect& r) const { Mat o(r.height, r.width); for (int i = 0; i < r.height; ++i) for (int j = 0; j < r.width; ++j) o.d[(size_t)i * r.width + j] = at(r.y + i, r.x + j); return o; }
    bool empty() const { return d.empty(); }
};
struct Matx33d { double val[9]; Matx33d() : val{} {} Matx33d(const Mat& m) { LVREF_CHECK(m.rows == 3 && m.cols == 3, "Matx33d from a Mat of another shape"); for (int k = 0; k < 9; ++k) val[k] = m.d[k]; } double operator()(int i, int j) const { return val[3 * i + j]; } };
struct Vec3d { double val[3]; Vec3d() : val{} {} Vec3d(const Mat& m) { LVREF_CHECK(m.rows * m.cols == 3, "Vec3d from a Mat of another shape"); for (int k = 0; k < 3; ++k) val[k] = m.d[k]; } double operator()(int i) const { return val[i]; } double operator[](int i) const { return val[i]; } };
inline void cv2eigen(const Matx33d& s, Eigen::XMat& d) { LVREF_CHECK(d.rows() == 3 && d.cols() == 3, "cv2eigen: 3x3"); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) d(i, j) = s(i, j); }
inline void cv2eigen(const Vec3d& s, Eigen::XMat& d) { LVREF_CHECK(d.size() == 3, "cv2eigen: 3-vector"); for (int i = 0; i < 3; ++i) d(i) = s(i); }
class FileNode {
public:
    bool present = false; std::string text; std::map<std::string, FileNode> kids; Mat mat;
    operator double() const { return present ? std::atof(text.c_str()) : 0.0; }
    operator float() const { return (float)(double)*this; }
    operator int() const { return present ? (int)std::lround(std::atof(text.c_str())) : 0; }
    operator std::string() const { return text; }
    FileNode operator[](const std::string& k) const { auto it = kids.find(k); return it == kids.end() ? FileNode() : it->second; }
    FileNode operator[](const char* k) const { return (*this)[std::string(k)]; }
    bool empty() const { return !present; }
};
inline void operator>>(const FileNode& n, Mat& m) { m = n.mat; }
inline void operator>>(const FileNode& n, std::string& s) { s = n.text; }
inline void operator>>(const FileNode& n, double& v) { v = (double)n; }
inline void operator>>(const FileNode& n, int& v) { v = (int)n; }
class FileStorage {
    FileNode root; bool ok = false;
    static std::string trim(const std::string& s) { size_t a = 0, b = s.size(); while (a < b && std::isspace((unsigned char)s[a])) ++a; while (b > a && std::isspace((unsigned char)s[b - 1])) --b; return s.substr(a, b - a); }
    static std::string unquote(const std::string& s) { if (s.size() >= 2 && (s[0] == '"' || s[0] == '\'') && s.back() == s[0]) return s.substr(1, s.size() - 2); return s; }
public:
    enum { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string& path, int) { open(path, READ); }
    bool isOpened() const { return ok; }
    void release() {}
    bool open(const std::string& path, int)
    {
        std::ifstream f(path); if (!f) return false;
        std::vector<std::string> lines; std::string l;
        while (std::getline(f, l)) {      // a '#' outside quotes starts a comment
            char q = 0; for (size_t k = 0; k < l.size(); ++k) { if (q) { if (l[k] == q) q = 0; } else if (l[k] == '"' || l[k] == '\'') q = l[k]; else if (l[k] == '#') { l = l.substr(0, k); break; } }
            lines.push_back(l);
        }
        std::string parent;                                                   // the open nested map / matrix node, "" at top level
        for (size_t i = 0; i < lines.size(); ++i) {
            const std::string& raw = lines[i];
            if (trim(raw).empty() || raw[0] == '%' || trim(raw) == "---") continue;
            const bool indented = std::isspace((unsigned char)raw[0]);
            const size_t colon = raw.find(':');
            if (colon == std::string::npos) continue;
            const std::string key = trim(raw.substr(0, colon)); std::string val = trim(raw.substr(colon + 1));
            if (!indented) {
                parent.clear();
                FileNode n; n.present = true;
                if (val.rfind("!!opencv-matrix", 0) == 0) { parent = key; val.clear(); }
                else if (val.empty()) parent = key;
                n.text = unquote(val);
                root.kids[key] = n;
            } else if (!parent.empty()) {
                FileNode& p = root.kids[parent];
                if (key == "data") {
                    std::string all = val;                                     // [ ... ] possibly over several lines
                    while (all.find(']') == std::string::npos && i + 1 < lines.size()) all += " " + trim(lines[++i]);
                    for (char& ch : all) if (ch == '[' || ch == ']' || ch == ',') ch = ' ';
                    std::istringstream ss(all); double v; std::vector<double> vals; while (ss >> v) vals.push_back(v);
                    const int r = (int)p.kids["rows"], c = (int)p.kids["cols"];
                    LVREF_CHECK((int)vals.size() == r * c, "opencv-matrix: data length");
                    p.mat = Mat(r, c); p.mat.d = vals;
                } else { FileNode n; n.present = true; n.text = unquote(val); p.kids[key] = n; }
            }
        }
        ok = true; return true;
    }
    FileNode operator[](const std::string& k) const { return root[k]; }
    FileNode operator[](const char* k) const { return root[std::string(k)]; }
};
}
