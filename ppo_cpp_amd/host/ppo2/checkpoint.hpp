// checkpoint.hpp -- reader / writer for the checkpoint files the reference produces through the graph's Saver
// (ppo2/ppo2.hpp:107-223, saver_def G:33460-33467): a TensorFlow "bundle V2" pair
//     <prefix>.index                  LevelDB-style table: key = tensor name, value = BundleEntryProto
//     <prefix>.data-00000-of-00001    raw little-endian fp32 tensors, concatenated in sorted-name order
// plus the JSON side-car <prefix>.json with hyper-parameters and the running statistics.  Written from the published
// file formats (no TensorFlow code): SSTable blocks with prefix-compressed keys and a restart array, 5-byte block
// trailers (type + masked CRC32C), a 48-byte footer with the table magic; BundleHeaderProto in the "" key.
// The writer reproduces the reference's own checkpoint byte for byte (tests/test_checkpoint.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ckpt {

struct Tensor { std::vector<int64_t> shape; std::vector<float> data; };
typedef std::map<std::string, Tensor> Bundle;          // sorted by name, like the table

// ---- CRC32C (Castagnoli), masked the LevelDB way -------------------------------------------------------------
inline uint32_t crc32c(const uint8_t* p, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1; table[i] = c; }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
inline uint32_t mask_crc(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xa282ead8u; }

// ---- varints / little-endian helpers ----------------------------------------------------------------------------
inline void put_varint(std::string& s, uint64_t v) { while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; } s.push_back((char)v); }
inline void put_fixed32(std::string& s, uint32_t v) { for (int i = 0; i < 4; ++i) s.push_back((char)((v >> (8 * i)) & 0xFF)); }
inline uint64_t get_varint(const std::string& s, size_t& p) {
    uint64_t r = 0; int shift = 0;
    for (;;) { if (p >= s.size()) throw std::runtime_error("ckpt: truncated varint"); const uint8_t b = (uint8_t)s[p++]; r |= (uint64_t)(b & 0x7F) << shift; if (!(b & 0x80)) return r; shift += 7; }
}
inline uint32_t get_fixed32(const std::string& s, size_t p) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)(uint8_t)s[p + i] << (8 * i); return v; }

// ---- table blocks -----------------------------------------------------------------------------------------------
class BlockBuilder {
public:
    explicit BlockBuilder(int restart_interval = 16) : interval_(restart_interval), counter_(0) { restarts_.push_back(0); }
    void add(const std::string& key, const std::string& value) {
        size_t shared = 0;
        if (counter_ < interval_) { const size_t n = std::min(last_.size(), key.size()); while (shared < n && last_[shared] == key[shared]) ++shared; }
        else { restarts_.push_back((uint32_t)buf_.size()); counter_ = 0; }
        put_varint(buf_, shared); put_varint(buf_, key.size() - shared); put_varint(buf_, value.size());
        buf_.append(key, shared, std::string::npos); buf_.append(value);
        last_ = key; ++counter_;
    }
    std::string finish() { std::string out = buf_; for (uint32_t r : restarts_) put_fixed32(out, r); put_fixed32(out, (uint32_t)restarts_.size()); return out; }
private:
    int interval_, counter_;
    std::string buf_, last_;
    std::vector<uint32_t> restarts_;
};

inline void write_block(std::string& file, const std::string& block, uint64_t* off, uint64_t* size) {
    *off = file.size(); *size = block.size();
    file.append(block);
    std::string tail(1, '\0');                                            // kNoCompression
    uint32_t crc = crc32c((const uint8_t*)block.data(), block.size());
    crc = crc32c((const uint8_t*)tail.data(), 1, crc);
    file.append(tail); put_fixed32(file, mask_crc(crc));
}

inline std::string entry_proto(const Tensor& t, uint64_t offset) {           // BundleEntryProto
    std::string e;
    e.push_back(0x08); put_varint(e, 1);                                      // dtype = DT_FLOAT
    std::string shape;
    for (int64_t d : t.shape) { std::string dim; dim.push_back(0x08); put_varint(dim, (uint64_t)d); shape.push_back(0x12); put_varint(shape, dim.size()); shape.append(dim); }
    e.push_back(0x12); put_varint(e, shape.size()); e.append(shape);
    if (offset) { e.push_back(0x20); put_varint(e, offset); }
    const uint64_t bytes = t.data.size() * sizeof(float);
    e.push_back(0x28); put_varint(e, bytes);
    e.push_back(0x35); put_fixed32(e, mask_crc(crc32c((const uint8_t*)t.data.data(), bytes)));
    return e;
}

inline void save_bundle(const std::string& prefix, const Bundle& b) {
    std::string data;
    BlockBuilder blk;
    blk.add("", std::string("\x08\x01\x1a\x02\x08\x01", 6));                 // BundleHeaderProto{num_shards 1, version{producer 1}}
    std::string last_key;
    for (const auto& kv : b) {
        blk.add(kv.first, entry_proto(kv.second, data.size()));
        data.append((const char*)kv.second.data.data(), kv.second.data.size() * sizeof(float));
        last_key = kv.first;
    }
    std::string file;
    uint64_t doff, dsize, moff, msize, ioff, isize;
    write_block(file, blk.finish(), &doff, &dsize);
    BlockBuilder meta; write_block(file, meta.finish(), &moff, &msize);
    std::string sep = last_key.empty() ? std::string() : std::string(1, (char)(last_key[0] + 1));   // short successor of the last key
    std::string handle; put_varint(handle, doff); put_varint(handle, dsize);
    BlockBuilder index(1); index.add(sep, handle); write_block(file, index.finish(), &ioff, &isize);
    std::string footer; put_varint(footer, moff); put_varint(footer, msize); put_varint(footer, ioff); put_varint(footer, isize);
    footer.resize(40, '\0');
    const uint64_t magic = 0xdb4775248b80fb57ull;
    for (int i = 0; i < 8; ++i) footer.push_back((char)((magic >> (8 * i)) & 0xFF));
    file.append(footer);
    std::ofstream(prefix + ".index", std::ios::binary).write(file.data(), (std::streamsize)file.size());
    std::ofstream(prefix + ".data-00000-of-00001", std::ios::binary).write(data.data(), (std::streamsize)data.size());
}

inline std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("ckpt: cannot open " + path);
    return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

inline Bundle load_bundle(const std::string& prefix) {
    const std::string file = slurp(prefix + ".index"), data = slurp(prefix + ".data-00000-of-00001");
    if (file.size() < 48) throw std::runtime_error("ckpt: index too short");
    const std::string footer = file.substr(file.size() - 48);
    size_t p = 0;
    get_varint(footer, p); get_varint(footer, p);
    const uint64_t ioff = get_varint(footer, p), isize = get_varint(footer, p);
    auto check_block = [&](uint64_t off, uint64_t size) {
        if (off + size + 5 > file.size()) throw std::runtime_error("ckpt: block out of range");
        uint32_t crc = crc32c((const uint8_t*)file.data() + off, size + 1);
        if (mask_crc(crc) != get_fixed32(file, off + size + 1)) throw std::runtime_error("ckpt: block checksum mismatch");
        return file.substr(off, size);
    };
    auto entries = [&](const std::string& blk) {
        std::vector<std::pair<std::string, std::string>> out;
        const uint32_t nr = get_fixed32(blk, blk.size() - 4);
        const size_t end = blk.size() - 4 - 4 * (size_t)nr;
        size_t q = 0; std::string key;
        while (q < end) {
            const uint64_t shared = get_varint(blk, q), non_shared = get_varint(blk, q), vlen = get_varint(blk, q);
            key = key.substr(0, shared) + blk.substr(q, non_shared); q += non_shared;
            out.push_back({key, blk.substr(q, vlen)}); q += vlen;
        }
        return out;
    };
    Bundle b;
    for (const auto& ih : entries(check_block(ioff, isize))) {
        size_t q = 0;
        const uint64_t off = get_varint(ih.second, q), size = get_varint(ih.second, q);
        for (const auto& kv : entries(check_block(off, size))) {
            if (kv.first.empty()) continue;                                   // header
            Tensor t; uint64_t toff = 0, tsize = 0; uint32_t tcrc = 0; bool has_crc = false;
            const std::string& v = kv.second; size_t r = 0;
            while (r < v.size()) {
                const uint64_t tag = get_varint(v, r);
                const int field = (int)(tag >> 3), wire = (int)(tag & 7);
                if (wire == 0) { const uint64_t x = get_varint(v, r); if (field == 4) toff = x; else if (field == 5) tsize = x; else if (field == 1 && x != 1) throw std::runtime_error("ckpt: only DT_FLOAT tensors supported"); }
                else if (wire == 5) { tcrc = get_fixed32(v, r); has_crc = true; r += 4; }
                else if (wire == 2) {
                    const uint64_t len = get_varint(v, r);
                    if (field == 2) {
                        const std::string sh = v.substr(r, len); size_t s = 0;
                        while (s < sh.size()) { get_varint(sh, s); const uint64_t dl = get_varint(sh, s); size_t d = s; s += dl; while (d < s) { const uint64_t t2 = get_varint(sh, d); const uint64_t val = get_varint(sh, d); if ((t2 >> 3) == 1) t.shape.push_back((int64_t)val); } }
                    }
                    r += len;
                } else throw std::runtime_error("ckpt: unexpected wire type");
            }
            if (toff + tsize > data.size() || tsize % 4) throw std::runtime_error("ckpt: tensor out of range: " + kv.first);
            if (has_crc && mask_crc(crc32c((const uint8_t*)data.data() + toff, tsize)) != tcrc) throw std::runtime_error("ckpt: tensor checksum mismatch: " + kv.first);
            t.data.resize(tsize / 4);
            std::memcpy(t.data.data(), data.data() + toff, tsize);
            b[kv.first] = t;
        }
    }
    return b;
}

}  // namespace ckpt
