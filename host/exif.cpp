#include "exif.hpp"
#include <cstdio>
#include <cstring>

namespace mis {
namespace {
struct Tiff {
    const uint8_t* p; size_t n; bool le;
    bool ok(size_t o, size_t len) const { return o <= n && len <= n - o; }
    unsigned u16(size_t o) const { return le ? (unsigned)p[o] | ((unsigned)p[o + 1] << 8) : ((unsigned)p[o] << 8) | (unsigned)p[o + 1]; }
    unsigned long u32(size_t o) const {
        return le ? (unsigned long)p[o] | ((unsigned long)p[o + 1] << 8) | ((unsigned long)p[o + 2] << 16) | ((unsigned long)p[o + 3] << 24)
                  : ((unsigned long)p[o] << 24) | ((unsigned long)p[o + 1] << 16) | ((unsigned long)p[o + 2] << 8) | (unsigned long)p[o + 3];
    }
};
constexpr unsigned TAG_IMAGE_DESCRIPTION = 0x010E, TAG_EXIF_IFD = 0x8769, TYPE_ASCII = 2;

// one IFD: the tag's value if present, the Exif sub-IFD pointer, the offset of the next IFD
void walk_ifd(const Tiff& t, size_t off, bool* found, std::string* out, size_t* exif_ifd, size_t* next) {
    *next = 0;
    if (!off || !t.ok(off, 2)) return;
    const unsigned cnt = t.u16(off);
    if (!t.ok(off + 2, (size_t)cnt * 12 + 4)) return;
    for (unsigned i = 0; i < cnt; i++) {
        const size_t e = off + 2 + (size_t)i * 12;
        const unsigned tag = t.u16(e), type = t.u16(e + 2);
        const unsigned long count = t.u32(e + 4);
        if (tag == TAG_EXIF_IFD && exif_ifd) *exif_ifd = (size_t)t.u32(e + 8);
        if (tag != TAG_IMAGE_DESCRIPTION || type != TYPE_ASCII) continue;
        const size_t vo = count <= 4 ? e + 8 : (size_t)t.u32(e + 8);     // values of up to 4 bytes live in the entry itself
        if (!t.ok(vo, count)) continue;
        size_t len = 0;
        while (len < count && t.p[vo + len]) len++;                      // ASCII values are NUL terminated
        if (len > 1022) len = 1022;                                      // exif_entry_get_value(ee, buf, 1023): strncpy of maxlen - 1
        out->assign(reinterpret_cast<const char*>(t.p + vo), len);
        *found = true;
    }
    *next = (size_t)t.u32(off + 2 + (size_t)cnt * 12);
}
}  // namespace

bool exifImageDescription(const uint8_t* d, size_t n, std::string* out) {
    if (!d || !out || n < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;      // SOI
    size_t o = 2;
    while (o + 4 <= n) {
        if (d[o] != 0xFF) return false;
        const unsigned m = d[o + 1];
        if (m == 0xFF) { o++; continue; }                                        // fill byte
        if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { o += 2; continue; }   // no length
        if (m == 0xD9 || m == 0xDA) return false;                                // EOI / start of scan: no Exif segment before the image data
        const size_t len = ((size_t)d[o + 2] << 8) | d[o + 3];
        if (len < 2 || o + 2 + len > n) return false;
        if (m == 0xE1 && len >= 8 + 8 && !std::memcmp(d + o + 4, "Exif\0\0", 6)) {
            Tiff t{d + o + 10, len - 8, false};
            if (t.n < 8) return false;
            if (t.p[0] == 'I' && t.p[1] == 'I') t.le = true;
            else if (t.p[0] == 'M' && t.p[1] == 'M') t.le = false;
            else return false;
            if (t.u16(2) != 42) return false;
            bool found = false;
            size_t exif_ifd = 0, next = 0, unused = 0;
            walk_ifd(t, (size_t)t.u32(4), &found, out, &exif_ifd, &next);        // IFD0
            walk_ifd(t, next, &found, out, nullptr, &unused);                    // IFD1
            walk_ifd(t, exif_ifd, &found, out, nullptr, &unused);                // Exif sub-IFD
            return found;
        }
        o += 2 + len;
    }
    return false;
}

bool exifImageDescriptionFile(const std::string& path, std::string* out) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    // the Exif segment sits in front of the image data: the first 256 KB hold every APPn segment a camera writes (each <= 64 KB)
    std::vector<uint8_t> buf(256 * 1024);
    const size_t n = std::fread(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    return exifImageDescription(buf.data(), n, out);
}
}  // namespace mis
