// lvk_config.hpp — reads a LARVIO configuration file (the OpenCV-FileStorage YAML 1.0 subset of config/euroc.yaml and
// config/mynteye.yaml in the reference tree) into the two plain structs of include/lvk_c.h.  Replaces what
// ImageProcessor::loadParameters (/root/reference/src/image_processor.cpp:44-113) and LarVio::loadParameters
// (/root/reference/src/larvio.cpp:58-311) do with cv::FileStorage, without OpenCV: a maintainer keeps passing the same
// config file path.  Host-only and header-only (no GPU, no liblvk_hip symbols), so it can be unit-tested on any machine.
//
// Grammar accepted: "%YAML:1.0" / "---" header lines, '#' comments, top-level "key: scalar", one level of nested maps
// ("intrinsics:" followed by indented "fx: 458.654" → key "intrinsics.fx"), flow sequences "[a, b, …]" that may span lines, and
// "!!opencv-matrix" nodes (rows / cols / dt / data).  Like cv::FileNode, a missing numeric key reads as 0 and a missing string
// as ""; has() tells them apart.
#ifndef LVK_CONFIG_HPP
#define LVK_CONFIG_HPP
#include "lvk_c.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace lvk {

class ConfigFile {
public:
    struct Matrix { int rows, cols; std::vector<double> data; Matrix() : rows(0), cols(0) {} };

    bool open(const std::string& path)
    {
        FILE* f = std::fopen(path.c_str(), "rb");
        if (!f) { err_ = "cannot open " + path; return false; }
        std::string text; char buf[4096]; size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
        std::fclose(f);
        return parse(text);
    }

    bool parse(const std::string& text)
    {
        scalars_.clear(); seqs_.clear(); err_.clear();
        std::string parent;                    // the open nested map / matrix node ("" at top level)
        std::string seq_key, seq_text;         // a flow sequence still waiting for its ']'
        size_t pos = 0; int line_no = 0;
        while (pos <= text.size()) {
            size_t eol = text.find('\n', pos);
            if (eol == std::string::npos) eol = text.size();
            std::string line = strip_comment(text.substr(pos, eol - pos));
            pos = eol + 1; ++line_no;
            if (!seq_key.empty()) {            // continuation of "[ … ]"
                seq_text += " " + line;
                if (line.find(']') != std::string::npos) { if (!store_seq(seq_key, seq_text, line_no)) return false; seq_key.clear(); seq_text.clear(); }
                continue;
            }
            const size_t first = line.find_first_not_of(" \t\r");
            if (first == std::string::npos) continue;
            if (line[first] == '%' || line.compare(first, 3, "---") == 0 || line.compare(first, 3, "...") == 0) continue;
            const size_t colon = find_colon(line, first);
            if (colon == std::string::npos) {
                if (line[first] == '[' && !parent.empty()) {           // "data:" on one line, the bracket on the next
                    seq_key = parent + ".data"; seq_text = line.substr(first);
                    if (line.find(']') != std::string::npos) { if (!store_seq(seq_key, seq_text, line_no)) return false; seq_key.clear(); seq_text.clear(); }
                    continue;
                }
                return fail(line_no, "expected 'key: value'");
            }
            std::string key = trim(line.substr(first, colon - first));
            std::string val = trim(line.substr(colon + 1));
            if (first == 0) parent.clear(); else if (!parent.empty()) key = parent + "." + key;
            if (val.empty() || val.compare(0, 2, "!!") == 0) {        // a nested map or a tagged node opens
                if (first == 0) parent = key;
                else if (key.size() > 5 && key.compare(key.size() - 5, 5, ".data") == 0) { /* "data:" with the bracket on the next line */ }
                scalars_[key] = val;
                continue;
            }
            if (val[0] == '[') {
                if (val.find(']') != std::string::npos) { if (!store_seq(key, val, line_no)) return false; }
                else { seq_key = key; seq_text = val; }
                continue;
            }
            scalars_[key] = unquote(val);
        }
        if (!seq_key.empty()) return fail(line_no, "unterminated '[' of " + seq_key);
        return true;
    }

    bool has(const std::string& key) const { return scalars_.count(key) || seqs_.count(key); }
    // cv::FileNode → double / int / string conversions (missing → 0 / ""; a real read as int is rounded like cvRound)
    double real(const std::string& key) const
    {
        std::map<std::string, std::string>::const_iterator it = scalars_.find(key);
        return it == scalars_.end() ? 0.0 : std::strtod(it->second.c_str(), nullptr);
    }
    int integer(const std::string& key) const { return (int)std::lrint(real(key)); }
    std::string str(const std::string& key) const
    {
        std::map<std::string, std::string>::const_iterator it = scalars_.find(key);
        return it == scalars_.end() ? std::string() : it->second;
    }
    const std::vector<double>* seq(const std::string& key) const
    {
        std::map<std::string, std::vector<double> >::const_iterator it = seqs_.find(key);
        return it == seqs_.end() ? nullptr : &it->second;
    }
    // an !!opencv-matrix node (dt must be a numeric scalar type: d, f, i …; the data are kept as doubles, row-major)
    bool matrix(const std::string& key, Matrix* out) const
    {
        const std::vector<double>* d = seq(key + ".data");
        if (!d) return false;
        out->rows = integer(key + ".rows"); out->cols = integer(key + ".cols"); out->data = *d;
        return out->rows > 0 && out->cols > 0 && (size_t)out->rows * out->cols == d->size();
    }
    const std::string& error() const { return err_; }

private:
    static std::string trim(const std::string& s)
    {
        const size_t a = s.find_first_not_of(" \t\r"); if (a == std::string::npos) return std::string();
        const size_t b = s.find_last_not_of(" \t\r"); return s.substr(a, b - a + 1);
    }
    static std::string unquote(const std::string& s)
    {
        if (s.size() >= 2 && (s[0] == '"' || s[0] == '\'') && s[s.size() - 1] == s[0]) return s.substr(1, s.size() - 2);
        return s;
    }
    static std::string strip_comment(const std::string& s)
    {
        char q = 0;
        for (size_t i = 0; i < s.size(); ++i) {
            if (q) { if (s[i] == q) q = 0; }
            else if (s[i] == '"' || s[i] == '\'') q = s[i];
            else if (s[i] == '#' && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
        }
        return s;
    }
    static size_t find_colon(const std::string& s, size_t from)
    {   // the key/value separator: the first ':' outside quotes followed by blank or end of line
        char q = 0;
        for (size_t i = from; i < s.size(); ++i) {
            if (q) { if (s[i] == q) q = 0; }
            else if (s[i] == '"' || s[i] == '\'') q = s[i];
            else if (s[i] == '[') return std::string::npos;
            else if (s[i] == ':' && (i + 1 == s.size() || s[i + 1] == ' ' || s[i + 1] == '\t' || s[i + 1] == '\r')) return i;
        }
        return std::string::npos;
    }
    bool store_seq(const std::string& key, const std::string& text, int line_no)
    {
        const size_t a = text.find('['), b = text.rfind(']');
        if (a == std::string::npos || b == std::string::npos || b < a) return fail(line_no, "malformed sequence of " + key);
        std::vector<double> v; const char* p = text.c_str() + a + 1; const char* end = text.c_str() + b;
        while (p < end) {
            while (p < end && (*p == ' ' || *p == '\t' || *p == ',' || *p == '\r')) ++p;
            if (p >= end) break;
            char* q = nullptr; const double x = std::strtod(p, &q);
            if (q == p) return fail(line_no, "non-numeric element in sequence of " + key);
            v.push_back(x); p = q;
        }
        seqs_[key] = v;
        return true;
    }
    bool fail(int line_no, const std::string& what)
    {
        char b[32]; std::snprintf(b, sizeof b, "line %d: ", line_no); err_ = b + what; return false;
    }
    std::map<std::string, std::string> scalars_;
    std::map<std::string, std::vector<double> > seqs_;
    std::string err_;
};

// image_processor.cpp:44-113.  false (and *err) when the file cannot be read or a value is outside what the library implements.
inline bool load_fe_config(const ConfigFile& f, lvk_fe_config* c, std::string* err)
{
    std::memset(c, 0, sizeof *c);
    c->width = f.integer("resolution_width"); c->height = f.integer("resolution_height");            // :74-76
    c->pyramid_levels = f.integer("pyramid_levels"); c->patch_size = f.integer("patch_size");           // :52-53
    c->max_iteration = f.integer("max_iteration"); c->track_precision = f.real("track_precision");     // :54-55
    c->max_features_num = f.integer("max_features_num"); c->min_distance = f.integer("min_distance");  // :58-59
    c->flag_equalize = f.integer("flag_equalize") ? 1 : 0; c->pub_frequency = f.integer("pub_frequency");  // :60-62
    const std::string model = f.str("distortion_model");                                                 // :72
    if (model == "radtan") c->distortion_model = 0;
    else if (model == "equidistant") c->distortion_model = 1;
    else { if (err) *err = "distortion_model '" + model + "' (radtan and equidistant are supported)"; return false; }
    static const char* ik[4] = {"intrinsics.fx", "intrinsics.fy", "intrinsics.cx", "intrinsics.cy"};    // :78-82
    static const char* dk[4] = {"distortion_coeffs.k1", "distortion_coeffs.k2", "distortion_coeffs.p1", "distortion_coeffs.p2"};   // :84-88
    for (int i = 0; i < 4; ++i) { c->intrinsics[i] = f.real(ik[i]); c->distortion[i] = f.real(dk[i]); }
    ConfigFile::Matrix T;
    if (!f.matrix("T_cam_imu", &T) || T.rows != 4 || T.cols != 4) { if (err) *err = "T_cam_imu is not a 4x4 matrix"; return false; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c->R_cam_imu[3 * i + j] = T.data[4 * j + i];   // :90-93: the transpose
    return true;
}

// larvio.cpp:58-311
inline bool load_ekf_config(const ConfigFile& f, lvk_ekf_config* c, std::string* err)
{
    std::memset(c, 0, sizeof *c);
    c->if_fej = f.integer("if_FEJ") ? 1 : 0; c->estimate_extrin = f.integer("estimate_extrin") ? 1 : 0;
    c->estimate_td = f.integer("estimate_td") ? 1 : 0; c->if_zupt_valid = f.integer("if_ZUPT_valid") ? 1 : 0;
    c->sw_size = f.integer("sw_size"); c->max_track_len = f.integer("max_track_len");
    c->least_observation_number = f.integer("least_observation_number");
    c->max_features_in_one_grid = f.integer("max_features_in_one_grid");
    c->aug_grid_rows = f.integer("aug_grid_rows"); c->aug_grid_cols = f.integer("aug_grid_cols");
    c->pub_frequency = f.real("pub_frequency"); c->imu_rate = f.real("imu_rate");
    c->width = f.integer("resolution_width"); c->height = f.integer("resolution_height");
    static const char* ik[4] = {"intrinsics.fx", "intrinsics.fy", "intrinsics.cx", "intrinsics.cy"};
    for (int i = 0; i < 4; ++i) c->intrinsics[i] = f.real(ik[i]);
    ConfigFile::Matrix T;
    if (!f.matrix("T_cam_imu", &T) || T.rows != 4 || T.cols != 4) { if (err) *err = "T_cam_imu is not a 4x4 matrix"; return false; }
    for (int i = 0; i < 16; ++i) c->T_cam_imu[i] = T.data[i];
    c->td = f.real("td");
    c->noise_gyro = f.real("noise_gyro"); c->noise_acc = f.real("noise_acc");
    c->noise_gyro_bias = f.real("noise_gyro_bias"); c->noise_acc_bias = f.real("noise_acc_bias"); c->noise_feature = f.real("noise_feature");
    c->initial_covariance_orientation = f.real("initial_covariance_orientation");
    c->initial_covariance_velocity = f.real("initial_covariance_velocity");
    c->initial_covariance_position = f.real("initial_covariance_position");
    c->initial_covariance_gyro_bias = f.real("initial_covariance_gyro_bias");
    c->initial_covariance_acc_bias = f.real("initial_covariance_acc_bias");
    c->initial_covariance_extrin_rot = f.real("initial_covariance_extrin_rot");
    c->initial_covariance_extrin_trans = f.real("initial_covariance_extrin_trans");
    c->rotation_threshold = f.real("rotation_threshold"); c->translation_threshold = f.real("translation_threshold");
    c->tracking_rate_threshold = f.real("tracking_rate_threshold");
    c->feature_translation_threshold = f.real("feature_translation_threshold");
    c->zupt_max_feature_dis = f.real("zupt_max_feature_dis");
    c->zupt_noise_v = f.real("zupt_noise_v"); c->zupt_noise_p = f.real("zupt_noise_p"); c->zupt_noise_q = f.real("zupt_noise_q");
    c->static_duration = f.real("static_duration");
    c->feature_idp_dim = f.integer("feature_idp_dim"); c->use_schmidt = f.integer("use_schmidt") ? 1 : 0;
    c->calib_imu_instrinsic = f.integer("calib_imu_instrinsic") ? 1 : 0;
    c->max_features = f.integer("max_features_num");
    if (c->feature_idp_dim != 1) { if (err) *err = "feature_idp_dim must be 1 (the 3-D inverse-depth parametrisation is not implemented)"; return false; }
    if (c->use_schmidt) { if (err) *err = "use_schmidt must be 0 (the Schmidt variant is not implemented)"; return false; }
    return true;
}

}  // namespace lvk
#endif
