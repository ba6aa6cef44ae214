// hostcodec.hip -- host-only helpers for the Feather-in-zip wire format (stage a9, save_zip.py:30-100).
// pandas/pyarrow write Feather V2 with LZ4-FRAME compressed buffers by default and the GPU box image has no
// pyarrow, so himo_amd/feather.py reads/writes the Arrow IPC container itself and calls this decoder.
#include "himo_common.h"
#include <string.h>

namespace {

// LZ4 block format: token (literal length << 4 | match length - 4), literals, 2-byte offset, extended lengths
int64_t lz4_block(const unsigned char* src, int64_t n, unsigned char* dst, int64_t cap, int64_t dpos) {
    int64_t s = 0;
    while (s < n) {
        const unsigned tok = src[s++];
        int64_t lit = tok >> 4;
        if (lit == 15) { unsigned b; do { if (s >= n) return -1; b = src[s++]; lit += b; } while (b == 255); }
        if (s + lit > n || dpos + lit > cap) return -1;
        memcpy(dst + dpos, src + s, (size_t)lit);
        s += lit; dpos += lit;
        if (s >= n) break;                                   // last sequence has no match part
        if (s + 2 > n) return -1;
        const int64_t off = src[s] | (src[s + 1] << 8);
        s += 2;
        if (off == 0 || off > dpos) return -1;
        int64_t ml = (tok & 15);
        if (ml == 15) { unsigned b; do { if (s >= n) return -1; b = src[s++]; ml += b; } while (b == 255); }
        ml += 4;
        if (dpos + ml > cap) return -1;
        for (int64_t k = 0; k < ml; ++k) dst[dpos + k] = dst[dpos + k - off];   // may overlap: byte by byte
        dpos += ml;
    }
    return dpos;
}

}  // namespace

// LZ4 frame (magic 0x184D2204) -> dst; returns the decompressed size or -1 on malformed input / overflow.
// Block-independent and block-dependent frames are both handled (matches may reach into earlier blocks).
extern "C" int64_t himo_lz4_frame_decompress(const void* src_, int64_t n, void* dst_, int64_t cap) {
    const unsigned char* src = (const unsigned char*)src_;
    unsigned char* dst = (unsigned char*)dst_;
    if (!src || !dst || n < 7) return -1;
    if (!(src[0] == 0x04 && src[1] == 0x22 && src[2] == 0x4d && src[3] == 0x18)) return -1;
    const unsigned flg = src[4];
    if ((flg >> 6) != 1) return -1;                           // version
    const bool block_checksum = flg & 0x10, content_size = flg & 0x08, content_checksum = flg & 0x04, dict_id = flg & 0x01;
    int64_t s = 6 + (content_size ? 8 : 0) + (dict_id ? 4 : 0) + 1;   // FLG BD [size] [dict] HC
    int64_t dpos = 0;
    while (true) {
        if (s + 4 > n) return -1;
        const unsigned bs = src[s] | (src[s + 1] << 8) | (src[s + 2] << 16) | ((unsigned)src[s + 3] << 24);
        s += 4;
        if (bs == 0) break;                                   // end mark
        const bool raw = bs & 0x80000000u;
        const int64_t len = bs & 0x7fffffffu;
        if (s + len > n) return -1;
        if (raw) {
            if (dpos + len > cap) return -1;
            memcpy(dst + dpos, src + s, (size_t)len);
            dpos += len;
        } else {
            dpos = lz4_block(src + s, len, dst, cap, dpos);
            if (dpos < 0) return -1;
        }
        s += len + (block_checksum ? 4 : 0);
    }
    (void)content_checksum;
    return dpos;
}
