// Host check of csrc/codeobj_meta.h: parse the AMDGPU metadata note of a gfx950 code object hipcc produced and print, per kernel,
// where its hidden arguments live.  usage: codeobj_meta_test <file.co>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>

#include "../../tengine_amd/csrc/codeobj_meta.h"

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::map<std::string, tamd::HiddenLayout> m;
    if (!tamd::codeobj_hidden_layouts(buf.data(), buf.size(), m)) { printf("no metadata\n"); return 1; }
    for (auto& kv : m) {
        const tamd::HiddenLayout& h = kv.second;
        printf("%s explicit_end %d kernarg %d block_count %d %d %d group_size %d %d %d remainder %d %d %d grid_dims %d unknown_pointer %d\n",
               kv.first.c_str(), h.explicit_end, h.kernarg_size, h.block_count[0], h.block_count[1], h.block_count[2], h.group_size[0],
               h.group_size[1], h.group_size[2], h.remainder[0], h.remainder[1], h.remainder[2], h.grid_dims, (int)h.unknown_pointer);
    }
    // truncated / corrupted inputs must be refused, never read out of bounds
    for (size_t cut : {(size_t)10, (size_t)100, buf.size() / 2, buf.size() - 7}) {
        std::map<std::string, tamd::HiddenLayout> t;
        (void)tamd::codeobj_hidden_layouts(buf.data(), cut, t);
    }
    return 0;
}
