// gfw_jit.hip — per-clip specialisation of the fused frame kernel at run time.
//
// The reference compiles its OpenCL kernel per clip with the clip's constants substituted into the source
// (src/core/gpu/opencl.rs:181-214: lens model functions, pixel type, interpolation, flags folded to true/false).  The MI355X
// equivalent: gfw_frame.hip is embedded in the library as one amalgamated source text (tools/gen_jit_source.py, a build step) and
// compiled by hiprtc into ONE instantiation whose clip-invariant arguments (lens, sizes, strides, map constants, flags) are
// literals — GFW_BAKE_APPLY in the bake header gfw_api.hip generates.  Loads, uniform branches and the scalar registers they pin
// disappear: 78.9 -> 67.3 us per 4K C2 frame on MI355X, bit-identical output (profiles/r03_ab_bake.txt).
//
// The ahead-of-time kernels remain the product's floor: a context warps with them until the specialised kernel is ready, and for
// good if hiprtc is missing or the build fails (gfw_jit_status reports which).  Compilation runs on a worker thread (it needs no
// device); the module is loaded by the first launch that finds the code object ready.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <dirent.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gfw_jit.h"
#include "gfw_jit_source.inc"          // GFW_JIT_SOURCE (generated into the build directory)

namespace {

// hiprtc through dlopen: libgfwarp must load (and serve every frame ahead-of-time) on a box without it
typedef struct _hiprtcProgram *rtcProgram;
struct Rtc {
    void *lib = nullptr;
    int (*create)(rtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
    int (*compile)(rtcProgram, int, const char **) = nullptr;
    int (*log_size)(rtcProgram, size_t *) = nullptr;
    int (*log)(rtcProgram, char *) = nullptr;
    int (*code_size)(rtcProgram, size_t *) = nullptr;
    int (*code)(rtcProgram, char *) = nullptr;
    int (*destroy)(rtcProgram *) = nullptr;
    int (*version)(int *, int *) = nullptr;
    int ver_major = 0, ver_minor = 0;            // hiprtcVersion of the library that was found (0.0: none): two ROCm releases compile one source into kernels of different speed
    bool ok = false;
    Rtc() {
        if (const char *e = getenv("GFW_NO_HIPRTC")) { if (e[0] == '1') return; }        // tests: behave like a box without libhiprtc.so
        for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "libhiprtc.so.6", "/opt/rocm/lib/libhiprtc.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return;
#define GFW_RTC_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(lib, sym))
        GFW_RTC_SYM(create, "hiprtcCreateProgram"); GFW_RTC_SYM(compile, "hiprtcCompileProgram");
        GFW_RTC_SYM(log_size, "hiprtcGetProgramLogSize"); GFW_RTC_SYM(log, "hiprtcGetProgramLog");
        GFW_RTC_SYM(code_size, "hiprtcGetCodeSize"); GFW_RTC_SYM(code, "hiprtcGetCode"); GFW_RTC_SYM(destroy, "hiprtcDestroyProgram");
#undef GFW_RTC_SYM
        version = reinterpret_cast<decltype(version)>(dlsym(lib, "hiprtcVersion"));
        ok = create && compile && log_size && log && code_size && code && destroy;
        if (ok && version) (void)version(&ver_major, &ver_minor);
        // hiprtc loads its compiler (libamd_comgr: LLVM inside) and its built-in headers LAZILY, at the first compile — on a worker thread, AFTER the exit hook below has
        // been registered, so their exit handlers would run BEFORE the hook: a process that leaves main() while its first build is still compiling then tears LLVM
        // down under the worker and the hook waits for a thread that never returns (seen once in thirty runs of tests/cpp/test_multi_device beside other GPU
        // processes, where the build outlived a 36-frame clip: profiles/r06_exit_during_build.txt).  Loaded here, their handlers are older than the hook.
        if (ok) {
            for (const char *dep : {"libamd_comgr.so.3", "libamd_comgr.so.2", "libamd_comgr.so"}) if (dlopen(dep, RTLD_NOW | RTLD_GLOBAL)) break;
            for (const char *dep : {"libhiprtc-builtins.so.7", "libhiprtc-builtins.so"}) if (dlopen(dep, RTLD_NOW | RTLD_GLOBAL)) break;
        }
    }
};
void join_all_workers();
// The worker threads run inside hiprtc / comgr / LLVM.  Those libraries were dlopen'ed AFTER libgfwarp's own statics were constructed, so at process exit
// their globals are torn down BEFORE g_cache's destructor would join a still-running build (round-3 advisor finding: any context starts a ~0.3-1 s build
// after three frames, and gfw_destroy does not wait for it).  This hook is registered right after the dlopen — later than hiprtc's own exit handlers, so it
// runs before them — and joins every build in flight.
Rtc &rtc() {
    static Rtc r;
    static const bool hooked = (r.ok ? (void)atexit(join_all_workers) : (void)0, true);
    (void)hooked;
    return r;
}
// -fno-slp-vectorize: LLVM's SLP pass packs scalar f32 operations into v_pk_* forms fed by register shuffles; with the branch-free lane-row the same kernel
// runs 53.5 us per C2 frame with it and 46.2 without (profiles/r04_ab_fastrow.txt; packed ops issue in 4.7 cycles against 2 x 2.5, tools/microbench_mix.hip).
// A "-fslp-vectorize" entry of GFW_JIT_DEFS comes later on the command line and wins.
#define GFW_JIT_NO_SLP "-fno-slp-vectorize"

enum { ST_COMPILING = 1, ST_COMPILED = 2, ST_LOADED = 3, ST_FAILED = -1 };

// An entry of the definition list becomes -D<entry>; one that starts with '-' is a compiler option passed as it is (experiments through
// GFW_JIT_DEFS, e.g. "-fno-slp-vectorize": the options are part of the cache key like every definition).
std::string jit_option(const std::string &d) { return (!d.empty() && d[0] == '-') ? d : "-D" + d; }

struct Entry {
    std::atomic<int> state{ST_COMPILING};
    std::vector<char> code;
    std::string log;
    double compile_ms = 0.0;
    std::thread worker;
    std::mutex join_mu;                    // one joiner at a time; never held together with g_mu by a thread that waits for the build
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    void join() { std::lock_guard<std::mutex> lk(join_mu); if (worker.joinable()) worker.join(); }
    // At process exit only (join_all_workers): a build that has not come back within the hook's patience is left behind — the thread is detached and dies with the
    // process — instead of being waited for without end (see join_all_workers).
    // (try_lock: a context's thread may itself be waiting inside join() for this very build — GFW_OPT_JIT = 2 — when main() returns: the hook must not queue behind it)
    void abandon() { std::unique_lock<std::mutex> lk(join_mu, std::try_to_lock); if (lk.owns_lock() && worker.joinable()) worker.detach(); }
    ~Entry() {
        if (state.load(std::memory_order_acquire) == ST_COMPILING) abandon(); else join();
        if (worker.joinable()) (void)new std::thread(std::move(worker));       // could not be detached (see abandon): a joinable std::thread must never be destroyed
    }
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<Entry>> g_cache;       // key: device | arch | options | bake header
// How many specialisations a process keeps (a loaded module each: ~30 KB of code).  256 until round 6 — which the GPU test suite itself outgrew: run serially in
// one process (the way the driver runs it) it reached the cap two thirds of the way through, and the clips after that were served by the ahead-of-time kernels
// ("specialisation cache full"): correct pixels, but tests that ask for the specialised kernel by name failed (gpurun_out/r06_final4: 14 of 20 clips "served").
constexpr size_t kMaxEntries = 4096;

// ---- specialised kernels on disk --------------------------------------------------------------------------------------------------
// A code object is a pure function of (architecture, compiler options, bake header, embedded source): it can be kept.  Two directories are consulted before
// hiprtc is: <directory of libgfwarp.so>/jit_cache — shipped with the build, filled by tools/build_jit_cache.py for the BASELINE configurations, so that their
// specialised kernels need no libhiprtc.so at run time — and $GFW_JIT_CACHE, which also receives every kernel this process compiles (a render farm compiles a
// clip's kernel once).  File name: 128-bit FNV-1a of the key text, so a changed source, option or constant never finds a stale kernel.
static std::string key_hash(const std::string &key) {
    unsigned long long h1 = 1469598103934665603ull, h2 = 0x9ae16a3b2f90404full;
    for (unsigned char c : key) { h1 = (h1 ^ c) * 1099511628211ull; h2 = (h2 ^ (c + 0x9eu)) * 0x100000001b3ull; h2 ^= h2 >> 29; }
    char b[40]; snprintf(b, sizeof(b), "%016llx%016llx", h1, h2);
    return b;
}
static std::string lib_dir() {
    Dl_info di;
    if (!dladdr((const void *)&key_hash, &di) || !di.dli_fname) return std::string();
    std::string p = di.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
static std::string full_key(const std::string &arch, const std::vector<std::string> &opts, const std::string &header) {
    std::string key = arch;
    for (const std::string &o : opts) key += "|" + o;
    key += "|" + header + "|";
    key += std::to_string(sizeof(GFW_JIT_SOURCE)) + ":" + key_hash(GFW_JIT_SOURCE);          // the embedded source itself
    return key;
}
// A code object read back must at least be one: an ELF header and a plausible size (a file cut short by an interrupted writer used to be loaded, fail in
// hipModuleLoadData and mark the clip's kernel dead for the rest of the clip).
// hiprtc hands back a bare ELF today; a release that returns a clang offload bundle instead (plain "__CLANG_OFFLOAD_BUNDLE__" or compressed "CCOB" — hipModuleLoadData
// takes all three) must not silently turn the on-disk cache off: the three magics are accepted, and the first refusal is logged once.
static bool looks_like_code_object(const std::vector<char> &c) {
    if (c.size() <= 64) return false;
    if (c[0] == 0x7f && c[1] == 'E' && c[2] == 'L' && c[3] == 'F') return true;
    static const char bundle[] = "__CLANG_OFFLOAD_BUNDLE__";
    if (memcmp(c.data(), bundle, sizeof(bundle) - 1) == 0) return true;
    return c[0] == 'C' && c[1] == 'C' && c[2] == 'O' && c[3] == 'B';
}
static void log_refused_once(const char *what, const std::string &where) {
    static std::atomic<bool> said{false};
    if (!said.exchange(true)) fprintf(stderr, "[gfwarp] %s %s: not a code object (ELF / offload bundle) — the on-disk kernel cache ignores it\n", what, where.c_str());
}
static bool read_code_object(const std::string &path, std::vector<char> &code) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    bool ok = n > 0;
    if (ok) { code.resize((size_t)n); ok = fread(code.data(), 1, (size_t)n, f) == (size_t)n; }
    fclose(f);
    if (ok && !looks_like_code_object(code)) { log_refused_once("cached kernel", path); return false; }
    return ok;
}
// The shipped directory next to the library is part of THIS build (its kernels were compiled by the build's own compiler, like the ahead-of-time ones): its files
// are named by the key alone.  $GFW_JIT_CACHE may be shared across hosts and ROCm upgrades, and the compiler is part of what a kernel is (same source, ROCm 7.0
// against 7.2: 64 against 78 us per C2 bicubic frame, profiles/r04_ab_lut_rows.txt): its files carry the hiprtc version that produced them,
// <hash>.rtc<major>.<minor>.co, and a process finds only its own compiler's.  A process WITHOUT hiprtc cannot compile at all and takes any version's kernel.
static std::string rtc_tag() { Rtc &R = rtc(); return R.ok ? ".rtc" + std::to_string(R.ver_major) + "." + std::to_string(R.ver_minor) : std::string(); }
static bool cache_load(const std::string &hash, std::vector<char> &code, std::string &from) {
    const std::string ld = lib_dir();
    if (!ld.empty()) {
        const std::string path = ld + "/jit_cache/" + hash + ".co";
        if (read_code_object(path, code)) { from = path; return true; }
    }
    const char *e = getenv("GFW_JIT_CACHE");
    if (!e || !*e) return false;
    const std::string tag = rtc_tag();
    if (!tag.empty()) {
        const std::string path = std::string(e) + "/" + hash + tag + ".co";
        if (read_code_object(path, code)) { from = path; return true; }
        return false;
    }
    if (DIR *d = opendir(e)) {                     // no compiler here: whichever version's kernel the farm left
        std::string found;
        while (struct dirent *ent = readdir(d)) {
            const std::string n = ent->d_name;
            if (n.size() > hash.size() + 3 && n.compare(0, hash.size(), hash) == 0 && n.compare(n.size() - 3, 3, ".co") == 0 && (found.empty() || n > found)) found = n;
        }
        closedir(d);
        if (!found.empty() && read_code_object(std::string(e) + "/" + found, code)) { from = std::string(e) + "/" + found; return true; }
    }
    return false;
}
static bool write_atomically(const std::string &path, const std::vector<char> &code) {
    const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    if (fclose(f) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) { remove(tmp.c_str()); return false; }
    return true;
}
static void cache_store(const std::string &hash, const std::vector<char> &code) {
    const char *e = getenv("GFW_JIT_CACHE");
    if (!e || !*e) return;
    if (!looks_like_code_object(code)) { log_refused_once("compiled kernel for", hash); return; }
    (void)write_atomically(std::string(e) + "/" + hash + rtc_tag() + ".co", code);
}

// The exit hook.  It waits for the builds in flight — but not without end.  The hook is older than the exit handlers of everything the compiler constructs lazily
// DURING a compile (LLVM's function-local statics and managed statics, first touched on the worker thread): those run before it, under the compile, and can leave
// the worker blocked for good on something that no longer exists.  Preloading the compiler's libraries (rtc()) took the libraries' own handlers out of that race and
// made the hang rare, not impossible: once in the round's last serial run of the GPU tier `tests/cpp/test_multi_device 4 36` printed "multi-device ok", left main()
// with its background build still compiling and sat in exit() until the 120 s watchdog (gpurun_out/r06_ab, profiles/r06_exit_during_build.txt).  So: a build gets
// GFW_EXIT_WAIT_MS (default 20 s; a build takes 0.3-1 s, a few seconds beside other processes) to come back; one that does not is detached and ends with the process.
void join_all_workers() {
    std::vector<std::shared_ptr<Entry>> all;
    { std::lock_guard<std::mutex> lk(g_mu); for (auto &kv : g_cache) all.push_back(kv.second); }
    const char *env = getenv("GFW_EXIT_WAIT_MS");
    const long patience_ms = (env && atol(env) >= 0) ? atol(env) : 20000;
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(patience_ms);
    for (auto &e : all) {
        while (e->state.load(std::memory_order_acquire) == ST_COMPILING && std::chrono::steady_clock::now() < deadline) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        if (e->state.load(std::memory_order_acquire) != ST_COMPILING) e->join();          // published its result: the thread is on its way out
        else e->abandon();
    }
}

void compile_entry(Entry *e, std::string source, std::vector<std::string> opts, std::string hash) {
    const auto t0 = std::chrono::steady_clock::now();
    Rtc &R = rtc();
    rtcProgram prog = nullptr;
    int rc = R.create(&prog, source.c_str(), "gfw_jit.hip", 0, nullptr, nullptr);
    if (rc == 0) {
        std::vector<const char *> o;
        for (const std::string &s : opts) o.push_back(s.c_str());
        rc = R.compile(prog, (int)o.size(), o.data());
        size_t ls = 0;
        if (R.log_size(prog, &ls) == 0 && ls > 1) { e->log.resize(ls); (void)R.log(prog, &e->log[0]); }
        size_t cs = 0;
        if (rc == 0 && R.code_size(prog, &cs) == 0 && cs > 0) { e->code.resize(cs); rc = R.code(prog, e->code.data()); }
        else if (rc == 0) rc = -1;
        (void)R.destroy(&prog);
    }
    e->compile_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc == 0 && !hash.empty()) cache_store(hash, e->code);
    if (rc != 0 && e->log.empty()) e->log = "hiprtc error " + std::to_string(rc);
    e->state.store(rc == 0 ? ST_COMPILED : ST_FAILED, std::memory_order_release);
}

}  // namespace

// a code object to a file, through a temporary name (tools/build_jit_cache.py via gfw_debug_jit_compile: an interrupted build leaves no truncated entry behind)
bool gfw_jit_write_code_object(const std::string &path, const std::vector<char> &code) { return looks_like_code_object(code) && write_atomically(path, code); }
std::string gfw_jit_source_id() { return std::to_string(sizeof(GFW_JIT_SOURCE)) + ":" + key_hash(GFW_JIT_SOURCE); }
bool gfw_jit_available() { return rtc().ok; }      // (kernels cached on disk are served without it: gfw_jit_get)
// The file name a specialised kernel has in the on-disk caches (tools/build_jit_cache.py fills the shipped one through gfw_debug_jit_key).
std::string gfw_jit_cache_name(const std::string &arch, const std::vector<std::string> &defines, const std::string &bake_header) {
    std::vector<std::string> opts = {"--offload-arch=" + arch, "-O3", "-std=c++17", "-ffp-contract=off", GFW_JIT_NO_SLP, "-Wno-pass-failed",
                                     "-Wno-cuda-compat", "-DGFW_JIT=1", "-DGFW_BAKE=1"};
    for (const std::string &d : defines) opts.push_back(jit_option(d));
    return key_hash(full_key(arch, opts, bake_header)) + ".co";
}

// State of the specialised kernel for (device, options, header): starts the build on first sight.  wait: block until it is decided.
// Returns the function once loaded on `device` (which must be the calling thread's current device), nullptr otherwise.
hipFunction_t gfw_jit_get(int device, const std::string &arch, const std::vector<std::string> &defines, const std::string &bake_header,
                          bool wait, GfwJitInfo *info) {
    if (info) { info->state = GFW_JIT_UNAVAILABLE; info->compile_ms = 0.0; info->log.clear(); }
    std::string key = std::to_string(device) + "|" + arch;
    for (const std::string &d : defines) key += "|" + d;
    key += "|" + bake_header;
    std::shared_ptr<Entry> e;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_cache.find(key);
        if (it == g_cache.end()) {
            // a process that walks through thousands of distinct clips keeps its first kMaxEntries specialisations (modules stay loaded: kernels of
            // any of them may be in flight); later clips run ahead of time
            if (g_cache.size() >= kMaxEntries) { if (info) { info->state = GFW_JIT_UNAVAILABLE; info->log = "specialisation cache full"; } return nullptr; }
            std::vector<std::string> opts = {"--offload-arch=" + arch, "-O3", "-std=c++17", "-ffp-contract=off", GFW_JIT_NO_SLP, "-Wno-pass-failed",
                                             "-Wno-cuda-compat", "-DGFW_JIT=1", "-DGFW_BAKE=1"};
            for (const std::string &d : defines) opts.push_back(jit_option(d));
            const std::string hash = key_hash(full_key(arch, opts, bake_header));
            e = std::make_shared<Entry>();
            std::string from;
            if (cache_load(hash, e->code, from)) {                      // shipped with the build, or compiled by an earlier process: no hiprtc needed
                e->log = "code object from " + from;
                e->state.store(ST_COMPILED, std::memory_order_release);
            } else if (!rtc().ok) {
                if (info) info->log = "libhiprtc.so not found and no cached kernel for this clip (" + hash + ".co)";
                return nullptr;
            } else {
                std::string source = bake_header + "\n" + GFW_JIT_SOURCE;
                e->worker = std::thread(compile_entry, e.get(), std::move(source), std::move(opts), hash);
            }
            g_cache.emplace(key, e);
        } else e = it->second;
    }
    if (wait && e->state.load(std::memory_order_acquire) == ST_COMPILING) e->join();      // outside g_mu: other contexts keep polling their own entries meanwhile
    int st = e->state.load(std::memory_order_acquire);
    if (st == ST_COMPILED) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (e->state.load() == ST_COMPILED) {
            e->join();                                   // the worker has published its result: returns at once
            hipError_t err = hipModuleLoadData(&e->mod, e->code.data());
            if (err == hipSuccess) err = hipModuleGetFunction(&e->fn, e->mod, "gfw_jit_kernel");
            // A specialised build whose lanes keep a kilobyte or more in scratch has its argument block there (2.2 KB, copied by every lane: 0.4 ms per launch
            // — the ahead-of-time kernels are several times faster): not used.  Nothing the shipped source compiles to does this; round 6 met it once, through a
            // pointer test in the kernel body that kept the optimiser from reading the argument segment directly (profiles/r06_radial_closed_form.txt).
            int scratch = 0;
            if (err == hipSuccess && hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, e->fn) == hipSuccess && scratch >= 1024) {
                char b[160]; snprintf(b, sizeof(b), "\nspecialised kernel refused: %d bytes of scratch per lane (its argument block lives there); the ahead-of-time kernel serves the clip", scratch);
                e->log += b; (void)hipModuleUnload(e->mod); e->mod = nullptr; e->fn = nullptr; e->state.store(ST_FAILED);
            }
            else if (err != hipSuccess) { e->log += std::string("\nmodule load: ") + hipGetErrorString(err); e->state.store(ST_FAILED); }
            else { e->code.clear(); e->code.shrink_to_fit(); e->state.store(ST_LOADED); }
        }
        st = e->state.load();
    }
    if (info) {
        info->state = st == ST_LOADED ? GFW_JIT_READY : st == ST_FAILED ? GFW_JIT_FAILED : GFW_JIT_COMPILING;
        if (st != ST_COMPILING) { std::lock_guard<std::mutex> lk(g_mu); info->compile_ms = e->compile_ms; info->log = e->log; }      // (a module-load error is appended under g_mu)
    }
    return st == ST_LOADED ? e->fn : nullptr;
}

// Build only (no device needed): the code object's size in bytes, or -1 with the compiler's log.  Host-side check of the embedded
// source and of the toolchain (tests/test_jit_host.py); `code_out` receives the code object when given.
long gfw_jit_compile_only(const std::string &arch, const std::vector<std::string> &defines, const std::string &bake_header, std::string &log,
                          std::vector<char> *code_out) {
    if (!rtc().ok) { log = "libhiprtc.so not found"; return -2; }
    Entry e;
    std::vector<std::string> opts = {"--offload-arch=" + arch, "-O3", "-std=c++17", "-ffp-contract=off", GFW_JIT_NO_SLP, "-Wno-pass-failed", "-Wno-cuda-compat",
                                     "-DGFW_JIT=1", "-DGFW_BAKE=1"};
    for (const std::string &d : defines) opts.push_back(jit_option(d));
    compile_entry(&e, bake_header + "\n" + GFW_JIT_SOURCE, opts, std::string());
    log = e.log;
    if (e.state.load() != ST_COMPILED) return -1;
    const long n = (long)e.code.size();
    if (code_out) code_out->swap(e.code);
    return n;
}

// Diagnosis builds (GFW_JIT_DEFS=GFW_TIMELINE=1): copy a __device__ array of the module that holds `fn` to the host.
bool gfw_jit_read_symbol(hipFunction_t fn, const char *name, void *dst, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_cache) {
        Entry *e = kv.second.get();
        if (e->fn != fn || !e->mod) continue;
        hipDeviceptr_t p = nullptr; size_t n = 0;
        if (hipModuleGetGlobal(&p, &n, e->mod, name) != hipSuccess || n < bytes) return false;
        return hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    }
    return false;
}

hipError_t gfw_jit_launch(hipFunction_t fn, const GfwClipArgs &C, int grid, hipStream_t s) {
    size_t size = sizeof(GfwClipArgs);
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<GfwClipArgs *>(&C), HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(fn, (unsigned)grid, 1, 1, 64, 4, 1, 0, s, nullptr, config);
}
