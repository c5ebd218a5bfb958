// How the library reads its environment.  Three classes, INTEGRATION.md section E lists every name:
//   * named runtime switches: getenv("TAMD_...") at the place of use -- user-facing behaviour (plan cache, direct dispatch, the
//     integer uint8 path, debug output, the fusion opt-outs read_tensor's error messages name);
//   * TAMD_PIN="key=value,key=value": test hook that pins a plan-time choice among LIVE candidates (tile configurations, kernel
//     forms the plan-time race would also consider) or switches a live optimisation off so that a test can compare with / without it.
//     Every key is set by a test (tests/test_abi.py checks the list against INTEGRATION.md).  Read at every prerun.
//   * experiment switches: exp_env("TAMD_...") is getenv only in builds with -DTAMD_EXPERIMENTS (tools/exp harnesses); in the
//     product it is a constant nullptr and the code behind it is dead.  Forms that lost their measured races live there.
#pragma once
#include <stdlib.h>
#include <string.h>

namespace tamd {

#ifdef TAMD_EXPERIMENTS
inline const char* exp_env(const char* name) { return getenv(name); }
#else
inline const char* exp_env(const char*) { return nullptr; }
#endif

// value of `key` in TAMD_PIN (nullptr: not pinned): a pointer INTO the environment string, so it stays valid while the variable is
// unchanged and several values can be held at once; the value ends at the next ',' or at the end of the string (callers parse
// numbers with atoi / sscanf, which stop there; a value never contains a comma: pwdw_cfg is written THxTWxthreads).
inline const char* tamd_pin(const char* key)
{
    const char* e = getenv("TAMD_PIN");
    if (!e) return nullptr;
    const size_t kl = strlen(key);
    for (const char* p = e; *p;) {
        const char* end = strchr(p, ',');
        const size_t len = end ? (size_t)(end - p) : strlen(p);
        if (len > kl && p[kl] == '=' && strncmp(p, key, kl) == 0) return p + kl + 1;
        p += len + (end ? 1 : 0);
    }
    return nullptr;
}
inline int tamd_pin_int(const char* key, int dflt) { const char* v = tamd_pin(key); return v ? atoi(v) : dflt; }

}  // namespace tamd
