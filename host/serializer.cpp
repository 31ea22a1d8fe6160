// serializer.cpp -- see serializer.hpp
#include "serializer.hpp"
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace mis {

std::vector<std::string> splitMatrixStrItems(std::string_view sv) {
    std::vector<std::string> ret;
    for (auto pos = sv.find(','); pos != sv.npos; pos = sv.find(',')) {
        ret.emplace_back(sv.substr(0, pos));
        sv = sv.substr(pos + 1);
    }
    ret.emplace_back(sv);
    return ret;
}

template <typename T>
static std::string serialize_impl(const T* m, int rows, int cols) {
    std::stringstream ss;   // default formatting, as the reference: precision 6, no fixed / scientific flag
    ss << "[";
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++) ss << m[r * cols + c] << (c == cols - 1 ? ";" : ",");
    ss << "]";
    return ss.str();
}
std::string serializeMatrix(const double* m, int rows, int cols) { return serialize_impl(m, rows, cols); }
std::string serializeMatrix(const float* m, int rows, int cols) { return serialize_impl(m, rows, cols); }

MatF deserializeMatrix(std::string s) {
    if (s.size() < 2 || s[0] != '[') throw std::runtime_error("matrix text must start with '['");
    s = s.substr(1);
    std::vector<double> values;
    int nCols = 0, nRows = 0;
    const char* data = s.c_str();
    while (true) {
        char* end = nullptr;
        values.push_back((double)std::strtold(data, &end));
        if (end == data || *end == '\0') throw std::runtime_error("malformed matrix text");
        data = end + 1;
        if (*end == ';') {
            if (nRows == 0) nCols++;
            nRows++;
        } else if (nRows == 0) {
            nCols++;
        }
        if (*data == ']') break;
        if (*data == '\0') throw std::runtime_error("matrix text must end with ']'");
    }
    MatF ret;
    ret.rows = nRows; ret.cols = nCols;
    ret.v.assign((size_t)nRows * nCols, 0.f);
    for (int i = 0; i < nRows; i++)
        for (int j = 0; j < nCols; j++) ret.v[(size_t)i * nCols + j] = (float)values[(size_t)nCols * i + j];
    return ret;
}

void serializeCameraParams(const std::vector<CameraParams>& cams, const std::string& path) {
    std::fstream fs;
    fs.open(path, std::ios::out);
    if (!fs) throw std::runtime_error("can't write " + path);
    for (const CameraParams& c : cams) {
        double R[9];
        for (int i = 0; i < 9; i++) R[i] = c.R(i / 3, i % 3);
        fs << c.aspect << "@" << c.focal << "@" << c.ppx << "@" << c.ppy << "@" << serializeMatrix(c.t.data(), 3, 1) << "@"
           << serializeMatrix(R, 3, 3) << std::endl;
    }
}

std::vector<CameraParams> deserializeCameraParams(const std::string& path) {
    std::vector<CameraParams> ret;
    std::fstream fs;
    fs.open(path, std::ios::in);
    if (!fs) throw std::runtime_error("can't read " + path);
    std::string line;
    while (std::getline(fs, line)) {
        if (line.empty()) continue;
        std::string f[6];
        for (int i = 0; i < 5; i++) {
            auto pos = line.find('@');
            if (pos == line.npos) throw std::runtime_error("cams.data: a line needs 6 '@'-separated fields");
            f[i] = line.substr(0, pos);
            line = line.substr(pos + 1);
        }
        f[5] = line;
        CameraParams c;
        c.aspect = std::strtod(f[0].c_str(), nullptr);
        c.focal = std::strtod(f[1].c_str(), nullptr);
        c.ppx = std::strtod(f[2].c_str(), nullptr);
        c.ppy = std::strtod(f[3].c_str(), nullptr);
        const MatF t = deserializeMatrix(f[4]), R = deserializeMatrix(f[5]);
        if (t.rows * t.cols != 3 || R.rows != 3 || R.cols != 3) throw std::runtime_error("cams.data: t must have 3 elements and R be 3x3");
        for (int i = 0; i < 3; i++) c.t[i] = t.v[i];
        for (int i = 0; i < 9; i++) c.R(i / 3, i % 3) = R.v[i];
        ret.emplace_back(c);
    }
    return ret;
}

void serializeIndices(const std::vector<int>& indices, const std::string& path) {
    std::fstream fs;
    fs.open(path, std::ios::out);
    if (!fs) throw std::runtime_error("can't write " + path);
    for (int i : indices) fs << i << std::endl;
}

std::vector<int> deserializeIndices(const std::string& path) {
    std::fstream fs;
    fs.open(path, std::ios::in);
    if (!fs) throw std::runtime_error("can't read " + path);
    std::vector<int> ret;
    std::string line;
    while (std::getline(fs, line))
        if (!line.empty()) ret.emplace_back((int)std::strtol(line.c_str(), nullptr, 10));
    return ret;
}

}  // namespace mis
