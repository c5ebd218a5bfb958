// Where a kernel's HIDDEN arguments live, read from the code object itself.
//
// direct.cc builds AQL kernel-dispatch packets by hand; the explicit arguments are recorded with the kernel's own parameter types
// (launch_rec.h), but behind them the compiler expects the "hidden" arguments HIP normally fills in (block counts, group sizes,
// grid dimensionality ...).  Their offsets are not an ABI constant -- they are listed per kernel in the code object's
// NT_AMDGPU_METADATA note (an ELF note named "AMDGPU", type 32, holding a MessagePack map: amdhsa.kernels -> [ { .symbol, .args ->
// [ { .offset, .size, .value_kind } ] } ]).  This header parses exactly that: ELF64 note walk + a minimal MessagePack reader.
// Plain C++, no HIP / HSA types: tests/csrc/codeobj_meta_test.cc runs it on the host against objects hipcc produces.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace tamd {

struct HiddenLayout {                // byte offsets inside the kernel-argument segment; -1: the kernel does not take that argument
    int block_count[3] = {-1, -1, -1};
    int group_size[3] = {-1, -1, -1};
    int remainder[3] = {-1, -1, -1};
    int grid_dims = -1;
    int global_offset[3] = {-1, -1, -1};
    int explicit_end = 0;            // end of the last explicit argument
    int kernarg_size = 0;            // .kernarg_segment_size
    bool unknown_pointer = false;    // a hidden argument this file does not know how to fill and that is a POINTER the kernel may
                                     // dereference (printf / hostcall / heap / queue ...): zero is what HIP passes when unused
};

namespace meta_detail {

struct Reader {
    const uint8_t* p; const uint8_t* end; bool ok = true;
    bool need(size_t n) { if ((size_t)(end - p) < n) { ok = false; return false; } return true; }
    uint64_t be(int n) { uint64_t v = 0; if (!need(n)) return 0; for (int i = 0; i < n; i++) v = (v << 8) | *p++; return v; }
};

enum Kind { K_NIL, K_BOOL, K_INT, K_STR, K_BIN, K_ARRAY, K_MAP, K_FLOAT, K_BAD };
struct Tok { Kind kind = K_BAD; int64_t i = 0; const char* s = nullptr; uint32_t n = 0; };     // n: string bytes / element count

// reads one token header; strings / bins are consumed, containers leave their elements to the caller
inline Tok next(Reader& r)
{
    Tok t;
    if (!r.need(1)) return t;
    const uint8_t b = *r.p++;
    auto str = [&](uint32_t n) { t.kind = K_STR; t.n = n; if (r.need(n)) { t.s = (const char*)r.p; r.p += n; } else t.kind = K_BAD; };
    if (b <= 0x7f) { t.kind = K_INT; t.i = b; }
    else if (b >= 0xe0) { t.kind = K_INT; t.i = (int8_t)b; }
    else if (b >= 0xa0 && b <= 0xbf) str(b & 0x1f);
    else if (b >= 0x90 && b <= 0x9f) { t.kind = K_ARRAY; t.n = b & 0x0f; }
    else if (b >= 0x80 && b <= 0x8f) { t.kind = K_MAP; t.n = b & 0x0f; }
    else switch (b) {
        case 0xc0: t.kind = K_NIL; break;
        case 0xc2: t.kind = K_BOOL; t.i = 0; break;
        case 0xc3: t.kind = K_BOOL; t.i = 1; break;
        case 0xc4: case 0xc5: case 0xc6: { const uint32_t n = (uint32_t)r.be(1 << (b - 0xc4)); t.kind = K_BIN; t.n = n; if (r.need(n)) r.p += n; else t.kind = K_BAD; break; }
        case 0xca: t.kind = K_FLOAT; (void)r.be(4); break;
        case 0xcb: t.kind = K_FLOAT; (void)r.be(8); break;
        case 0xcc: t.kind = K_INT; t.i = (int64_t)r.be(1); break;
        case 0xcd: t.kind = K_INT; t.i = (int64_t)r.be(2); break;
        case 0xce: t.kind = K_INT; t.i = (int64_t)r.be(4); break;
        case 0xcf: t.kind = K_INT; t.i = (int64_t)r.be(8); break;
        case 0xd0: t.kind = K_INT; t.i = (int8_t)r.be(1); break;
        case 0xd1: t.kind = K_INT; t.i = (int16_t)r.be(2); break;
        case 0xd2: t.kind = K_INT; t.i = (int32_t)r.be(4); break;
        case 0xd3: t.kind = K_INT; t.i = (int64_t)r.be(8); break;
        case 0xd9: str((uint32_t)r.be(1)); break;
        case 0xda: str((uint32_t)r.be(2)); break;
        case 0xdb: str((uint32_t)r.be(4)); break;
        case 0xdc: t.kind = K_ARRAY; t.n = (uint32_t)r.be(2); break;
        case 0xdd: t.kind = K_ARRAY; t.n = (uint32_t)r.be(4); break;
        case 0xde: t.kind = K_MAP; t.n = (uint32_t)r.be(2); break;
        case 0xdf: t.kind = K_MAP; t.n = (uint32_t)r.be(4); break;
        default: t.kind = K_BAD; break;           // ext types do not occur in this note
    }
    if (!r.ok) t.kind = K_BAD;
    return t;
}

inline void skip(Reader& r, const Tok& t, int depth = 0)
{
    if (depth > 32) { r.ok = false; return; }
    if (t.kind == K_ARRAY) for (uint32_t i = 0; i < t.n && r.ok; i++) { const Tok e = next(r); skip(r, e, depth + 1); }
    else if (t.kind == K_MAP) for (uint32_t i = 0; i < 2 * t.n && r.ok; i++) { const Tok e = next(r); skip(r, e, depth + 1); }
    else if (t.kind == K_BAD) r.ok = false;
}

inline bool is(const Tok& t, const char* s) { return t.kind == K_STR && t.n == strlen(s) && memcmp(t.s, s, t.n) == 0; }

inline void read_arg(Reader& r, const Tok& m, HiddenLayout& h)
{
    int64_t off = -1, size = 0;
    std::string kind;
    for (uint32_t i = 0; i < m.n && r.ok; i++) {
        const Tok k = next(r), v = next(r);
        if (is(k, ".offset") && v.kind == K_INT) off = v.i;
        else if (is(k, ".size") && v.kind == K_INT) size = v.i;
        else if (is(k, ".value_kind") && v.kind == K_STR) kind.assign(v.s, v.n);
        else skip(r, v);
    }
    if (off < 0) return;
    if (kind.compare(0, 7, "hidden_") != 0) { if (off + size > h.explicit_end) h.explicit_end = (int)(off + size); return; }
    static const char* axes = "xyz";
    for (int a = 0; a < 3; a++) {
        if (kind == std::string("hidden_block_count_") + axes[a]) { h.block_count[a] = (int)off; return; }
        if (kind == std::string("hidden_group_size_") + axes[a]) { h.group_size[a] = (int)off; return; }
        if (kind == std::string("hidden_remainder_") + axes[a]) { h.remainder[a] = (int)off; return; }
        if (kind == std::string("hidden_global_offset_") + axes[a]) { h.global_offset[a] = (int)off; return; }
    }
    if (kind == "hidden_grid_dims") { h.grid_dims = (int)off; return; }
    if (kind == "hidden_none") return;
    h.unknown_pointer = true;        // printf buffer, hostcall buffer, heap, default queue, completion action, multigrid sync ...
}

inline void read_kernel(Reader& r, const Tok& m, std::map<std::string, HiddenLayout>& out)
{
    HiddenLayout h;
    std::string symbol;
    for (uint32_t i = 0; i < m.n && r.ok; i++) {
        const Tok k = next(r), v = next(r);
        if (is(k, ".symbol") && v.kind == K_STR) symbol.assign(v.s, v.n);
        else if (is(k, ".kernarg_segment_size") && v.kind == K_INT) h.kernarg_size = (int)v.i;
        else if (is(k, ".args") && v.kind == K_ARRAY) {
            for (uint32_t a = 0; a < v.n && r.ok; a++) {
                const Tok am = next(r);
                if (am.kind == K_MAP) read_arg(r, am, h);
                else skip(r, am);
            }
        } else skip(r, v);
    }
    if (r.ok && !symbol.empty()) out[symbol] = h;
}

inline bool read_metadata(const uint8_t* p, size_t n, std::map<std::string, HiddenLayout>& out)
{
    Reader r{p, p + n};
    const Tok top = next(r);
    if (top.kind != K_MAP) return false;
    for (uint32_t i = 0; i < top.n && r.ok; i++) {
        const Tok k = next(r), v = next(r);
        if (is(k, "amdhsa.kernels") && v.kind == K_ARRAY) {
            for (uint32_t j = 0; j < v.n && r.ok; j++) {
                const Tok km = next(r);
                if (km.kind == K_MAP) read_kernel(r, km, out);
                else skip(r, km);
            }
        } else skip(r, v);
    }
    return r.ok;
}

}  // namespace meta_detail

// `elf`: one gfx code object (ELF64 little endian).  Adds "<symbol>.kd" -> layout for every kernel its metadata note lists.
// false: not an ELF64 object / no readable AMDGPU metadata note.
inline bool codeobj_hidden_layouts(const void* elf, size_t size, std::map<std::string, HiddenLayout>& out)
{
    const uint8_t* b = (const uint8_t*)elf;
    if (size < 64 || memcmp(b, "\177ELF", 4) != 0 || b[4] != 2 || b[5] != 1) return false;
    auto rd = [&](size_t off, int n) -> uint64_t { uint64_t v = 0; if (off + n > size) return 0; memcpy(&v, b + off, n); return v; };
    const uint64_t phoff = rd(32, 8);
    const unsigned phentsize = (unsigned)rd(54, 2), phnum = (unsigned)rd(56, 2);
    bool found = false;
    for (unsigned i = 0; i < phnum; i++) {
        const size_t ph = (size_t)phoff + (size_t)i * phentsize;
        if (ph + 56 > size) break;
        if ((uint32_t)rd(ph, 4) != 4) continue;                         // PT_NOTE
        const uint64_t off = rd(ph + 8, 8), filesz = rd(ph + 32, 8);
        if (off > size || filesz > size - off) continue;
        size_t q = (size_t)off;
        const size_t endq = (size_t)(off + filesz);
        while (q + 12 <= endq) {
            const uint32_t namesz = (uint32_t)rd(q, 4), descsz = (uint32_t)rd(q + 4, 4), type = (uint32_t)rd(q + 8, 4);
            const size_t name = q + 12, desc = name + ((namesz + 3) & ~3u);
            if (desc > endq || descsz > endq - desc) break;
            if (type == 32 && namesz == 7 && memcmp(b + name, "AMDGPU", 7) == 0)      // NT_AMDGPU_METADATA
                found = meta_detail::read_metadata(b + desc, descsz, out) || found;
            q = desc + ((descsz + 3) & ~3u);
        }
    }
    return found;
}

}  // namespace tamd
