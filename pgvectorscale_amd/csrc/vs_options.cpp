// vs_options.cpp — the tuning options of libvsgpu, behind the C ABI since round 6.
//
// Until round 5 every launch read some forty VS_* environment variables with getenv(): a drop-in library should take its options
// through its interface once.  Now an option is looked up in a table:
//   1. what the host set with vs_set_option(name, value)                       (wins; NULL unsets)
//   2. a SNAPSHOT of the process's VS_* environment variables — taken at the first lookup and again only when those variables
//      change (a fingerprint of the VS_* entries of `environ`, a few dozen pointer reads and three character compares per entry; no
//      getenv, no string search per option).  The environment stays what it was for the test and measurement tiers — a way to
//      flip a switch from outside the process — without being read on the launch path.
// Lookups are thread safe among themselves and against vs_set_option; changing the environment from another thread while a launch
// is being set up is as undefined as it was with getenv().
#include <cstdint>
#include <cstring>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>

#include "../../include/vsgpu.h"

extern char** environ;
void vs_set_error(const char* fmt, ...);

namespace {

struct Options {
    std::shared_mutex mu;
    std::unordered_map<std::string, std::string> set;  // vs_set_option
    std::unordered_map<std::string, std::string> env;  // the VS_* environment as of `fp`
    uint64_t fp = 0;
    bool have = false;
};
Options& opts() {
    static Options* o = new Options();  // (never destroyed: lookups may come from threads that outlive static destruction)
    return *o;
}

uint64_t env_fingerprint() {
    uint64_t h = 1469598103934665603ull;
    for (char** e = environ; e && *e; ++e) {
        const char* s = *e;
        if (s[0] != 'V' || s[1] != 'S' || s[2] != '_') continue;
        for (; *s; ++s) {
            h ^= (uint64_t)(unsigned char)*s;
            h *= 1099511628211ull;
        }
        h ^= 0xFFu;
        h *= 1099511628211ull;
    }
    return h;
}

void refresh_env(Options& o, uint64_t fp) {
    o.env.clear();
    for (char** e = environ; e && *e; ++e) {
        const char* s = *e;
        if (s[0] != 'V' || s[1] != 'S' || s[2] != '_') continue;
        const char* eq = strchr(s, '=');
        if (!eq) continue;
        o.env.emplace(std::string(s, (size_t)(eq - s)), std::string(eq + 1));
    }
    o.fp = fp;
    o.have = true;
}

}  // namespace

// the value of option `name`, or nullptr when it is set nowhere.  The pointer is this thread's: valid until its fourth lookup from now
// (a small ring of buffers — callers use a value at once; one that keeps it across other lookups copies it)
const char* vs_opt_get(const char* name) {
    static thread_local std::string ring[4];
    static thread_local unsigned turn = 0;
    std::string& buf = ring[turn++ & 3u];
    Options& o = opts();
    const uint64_t fp = env_fingerprint();
    {
        std::shared_lock<std::shared_mutex> lk(o.mu);
        if (o.have && o.fp == fp) {
            auto it = o.set.find(name);
            if (it == o.set.end()) {
                it = o.env.find(name);
                if (it == o.env.end()) return nullptr;
            }
            buf = it->second;
            return buf.c_str();
        }
    }
    std::unique_lock<std::shared_mutex> lk(o.mu);
    if (!o.have || o.fp != fp) refresh_env(o, fp);
    auto it = o.set.find(name);
    if (it == o.set.end()) {
        it = o.env.find(name);
        if (it == o.env.end()) return nullptr;
    }
    buf = it->second;
    return buf.c_str();
}

extern "C" int vs_set_option(const char* name, const char* value) {
    if (!name || strncmp(name, "VS_", 3) != 0 || strlen(name) > 64) {
        vs_set_error("vs_set_option: option names are the VS_* names of DESIGN.md section 10");
        return VS_ERR_INVALID;
    }
    try {
        Options& o = opts();
        std::unique_lock<std::shared_mutex> lk(o.mu);
        if (value) o.set[name] = value;
        else o.set.erase(name);
    } catch (const std::bad_alloc&) {
        vs_set_error("vs_set_option: out of host memory");
        return VS_ERR_OOM;
    }
    return VS_OK;
}

extern "C" int vs_get_option(const char* name, char* out, size_t cap) {
    if (!name || !out || cap == 0) {
        vs_set_error("vs_get_option: bad args");
        return VS_ERR_INVALID;
    }
    const char* v = vs_opt_get(name);
    if (!v) {
        out[0] = 0;
        return 0;  // not set anywhere
    }
    strncpy(out, v, cap - 1);
    out[cap - 1] = 0;
    return 1;
}
