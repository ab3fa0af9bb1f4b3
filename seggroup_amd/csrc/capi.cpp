// Error plumbing, version / device probe, and the label-file writers (a16 file side,
// reference seggroup/model.py:536-547) of libseggroup_hip.so.
#include <cerrno>
#include <string>

#include "sg_common.h"

namespace sg {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace sg

extern "C" {

const char* sg_last_error(void) { return sg::err_buf(); }

int sg_version(void) { return 100; }   // 0.1.0

int sg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// '%d\n' per value.  Values are small integers (labels, point ids < 2^31): a reverse-digit
// formatter into one contiguous buffer, one fwrite.
int sg_write_label_txt(const char* path, const int32_t* h_vec, int V) {
    if (!path || (V > 0 && !h_vec) || V < 0) return sg::fail(SG_EINVAL, "sg_write_label_txt: bad arguments");
    std::string buf;
    buf.resize((size_t)V * 12 + 1);
    char* o = &buf[0];
    for (int i = 0; i < V; ++i) {
        int64_t v = h_vec[i];
        if (v < 0) { *o++ = '-'; v = -v; }
        char tmp[12];
        int n = 0;
        do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
        while (n) *o++ = tmp[--n];
        *o++ = '\n';
    }
    FILE* f = fopen(path, "wb");
    if (!f) return sg::fail(SG_EINVAL, "sg_write_label_txt: cannot open %s: %s", path, strerror(errno));
    const size_t len = (size_t)(o - buf.data());
    const size_t w = fwrite(buf.data(), 1, len, f);
    if (fclose(f) != 0 || w != len) return sg::fail(SG_EINVAL, "sg_write_label_txt: short write to %s", path);
    return SG_OK;
}

// NumPy .npy v1.0, little-endian int32, C order, shape (V,)
int sg_write_label_npy(const char* path, const int32_t* h_vec, int V) {
    if (!path || (V > 0 && !h_vec) || V < 0) return sg::fail(SG_EINVAL, "sg_write_label_npy: bad arguments");
    char dict[128];
    int n = snprintf(dict, sizeof dict, "{'descr': '<i4', 'fortran_order': False, 'shape': (%d,), }", V);
    const int unpadded = 10 + n + 1;                        // magic(6)+ver(2)+len(2)+dict+'\n'
    const int pad = (64 - unpadded % 64) % 64;
    std::string hdr("\x93NUMPY\x01\x00", 8);
    const uint16_t hlen = (uint16_t)(n + pad + 1);
    hdr.push_back((char)(hlen & 0xff));
    hdr.push_back((char)(hlen >> 8));
    hdr.append(dict, n);
    hdr.append((size_t)pad, ' ');
    hdr.push_back('\n');
    FILE* f = fopen(path, "wb");
    if (!f) return sg::fail(SG_EINVAL, "sg_write_label_npy: cannot open %s: %s", path, strerror(errno));
    size_t w = fwrite(hdr.data(), 1, hdr.size(), f);
    w += fwrite(h_vec, 4, (size_t)V, f);
    if (fclose(f) != 0 || w != hdr.size() + (size_t)V) return sg::fail(SG_EINVAL, "sg_write_label_npy: short write to %s", path);
    return SG_OK;
}

}  // extern "C"
