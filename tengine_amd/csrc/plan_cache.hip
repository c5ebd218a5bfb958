// The plan cache (TAMD_PLAN_CACHE=<file>): what the plan-time autotune decided, shared by the three planners (graph_plan.hip,
// graph_u8.hip, graph_f32.hip).  Split out of graph.hip in round 6.
#include "graph.h"
#include "graph_internal.h"
#include "env.h"

#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <mutex>

#include "epilogue.h"

namespace tamd {

// ---- plan cache (TAMD_PLAN_CACHE=<file>): what the plan-time autotune decided, "<site>|<node>|<shape>" -> choice ---------------
// A first prerun measures as usual and writes the file; later preruns of the same model take the recorded choices WITHOUT
// launching anything -- a profiler then sees the run's own launches only (round 2's rocprofv3 CSVs were 99 % autotune
// dispatches), the plan no longer depends on one box's timing noise, and prerun drops from seconds to the packing time.
// The table is process-wide (graphs of one process share it) and guarded by a mutex; a file that changed on disk since it was
// read (size or modification time) is read again at the next lookup.
struct PlanCache {
    bool loaded = false, dirty = false;
    std::string path;
    long long stamp = 0;                                  // size ^ mtime of the file as read / written
    std::map<std::string, std::string> kv;     // the file's entries + this process's
    std::map<std::string, std::string> mine;   // what THIS process decided since the file was read (merged over the file at flush)
};
static std::mutex g_plan_cache_mu;
static long long file_stamp(const std::string& path)
{
    struct stat st;
    if (path.empty() || stat(path.c_str(), &st) != 0) return 0;
    return (long long)st.st_size * 1000003ll ^ (long long)st.st_mtim.tv_sec * 1000000007ll ^ (long long)st.st_mtim.tv_nsec;
}
// first line of a plan file: what the choices were made FOR.  A file written by another library version, for another
// architecture or with another candidate list is ignored as a whole (and overwritten at the next flush): a stale choice
// could name a configuration this build no longer launches
static std::string plan_cache_header()
{
    return std::string("#tamd-plan v2 gfx950 ") + tamd_version() + " gemm" + std::to_string(conv_igemm_num_cfgs()) + "/" + std::to_string(conv_pgemm_num_variants())
           + " u8" + std::to_string(conv_u8_gemm_num_cfgs()) + "/" + std::to_string(conv_u8_patch_num_cfgs());
}
static void plan_cache_read(const std::string& path, std::map<std::string, std::string>* kv)
{
    FILE* f = path.empty() ? nullptr : fopen(path.c_str(), "r");
    if (!f) return;
    char line[512];
    bool first = true, ok = false;
    while (fgets(line, sizeof(line), f)) {
        std::string l = line;
        while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
        if (first) { first = false; ok = l == plan_cache_header(); if (!ok) break; continue; }
        const size_t tab = l.find('\t');
        if (tab == std::string::npos) continue;
        (*kv)[l.substr(0, tab)] = l.substr(tab + 1);
    }
    fclose(f);
    if (!ok) kv->clear();
}
static PlanCache& plan_cache_locked()                     // call with g_plan_cache_mu held
{
    static PlanCache pc;
    const char* p = getenv("TAMD_PLAN_CACHE");
    const std::string want = p ? p : "";
    if (!pc.loaded || pc.path != want || (!pc.dirty && file_stamp(want) != pc.stamp)) {
        pc = PlanCache();
        pc.loaded = true; pc.path = want; pc.stamp = file_stamp(want);
        plan_cache_read(want, &pc.kv);
    }
    return pc;
}
bool plan_cache_get(const std::string& key, std::string* v)
{
    std::lock_guard<std::mutex> lk(g_plan_cache_mu);
    PlanCache& pc = plan_cache_locked();
    auto it = pc.kv.find(key);
    if (pc.path.empty() || it == pc.kv.end()) return false;
    *v = it->second;
    return true;
}
void plan_cache_put(const std::string& key, const std::string& v)
{
    std::lock_guard<std::mutex> lk(g_plan_cache_mu);
    PlanCache& pc = plan_cache_locked();
    if (pc.path.empty()) return;
    pc.kv[key] = v;
    pc.mine[key] = v;
    pc.dirty = true;
}
// Several processes may share one file (the ranks of a multi-GPU job): the entries on disk are merged with this process's own
// decisions (ours win), written to a temporary file and renamed over the old one -- a reader sees the old file or the new one,
// never half of either.
void plan_cache_flush()
{
    std::lock_guard<std::mutex> lk(g_plan_cache_mu);
    PlanCache& pc = plan_cache_locked();
    if (pc.path.empty() || !pc.dirty) return;
    std::map<std::string, std::string> merged;
    plan_cache_read(pc.path, &merged);
    for (auto& e : pc.mine) merged[e.first] = e.second;
    const std::string tmp = pc.path + ".tmp." + std::to_string((long)getpid());
    if (FILE* f = fopen(tmp.c_str(), "w")) {
        fprintf(f, "%s\n", plan_cache_header().c_str());
        for (auto& e : merged) fprintf(f, "%s\t%s\n", e.first.c_str(), e.second.c_str());
        fclose(f);
        if (rename(tmp.c_str(), pc.path.c_str()) != 0) (void)remove(tmp.c_str());
    }
    pc.kv = merged;
    pc.dirty = false;
    pc.stamp = file_stamp(pc.path);
}

}  // namespace tamd
