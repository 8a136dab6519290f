// aie_jit.h -- run-time specialisation of the gather-trade-build step / reset kernels (aie_specialize, include/aie.h).
//
// The build ships compile-time instances for the BASELINE configurations (aie_spec_generated.h): the parameter block
// folded into the code, 36 -> 28 us per C2 launch.  Any OTHER configuration gets the same treatment here, on request:
// the environment's normalised parameter block becomes the constant image of a translation unit that includes
// aie_kernels.hip (the very sources this library was built from, found beside the .so), hiprtc compiles it for the
// device's architecture (~5 s), and the code object is cached under ~/.cache/ai_economist_amd keyed by a hash of the
// image and the sources.  Everything here is optional: no hiprtc, no sources or no headers -> AIE_E_UNSUPPORTED and
// the environment keeps running the generic kernel.
#pragma once
#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace aie_jit {

typedef struct _hiprtcProgram* hiprtcProgram;
struct Rtc {
  void* lib = nullptr;
  int (*CreateProgram)(hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*CompileProgram)(hiprtcProgram, int, const char**) = nullptr;
  int (*GetProgramLogSize)(hiprtcProgram, size_t*) = nullptr;
  int (*GetProgramLog)(hiprtcProgram, char*) = nullptr;
  int (*GetCodeSize)(hiprtcProgram, size_t*) = nullptr;
  int (*GetCode)(hiprtcProgram, char*) = nullptr;
  int (*DestroyProgram)(hiprtcProgram*) = nullptr;
  int (*Version)(int*, int*) = nullptr;
  std::string path;  // where libhiprtc.so was found (its ROCm root holds the HIP and clang headers)
};

static inline bool load_rtc(Rtc& r) {
  for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
    r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.lib) break;
  }
  if (!r.lib) return false;
#define AIE_RTC_SYM(f) *reinterpret_cast<void**>(&r.f) = dlsym(r.lib, "hiprtc" #f)
  AIE_RTC_SYM(CreateProgram); AIE_RTC_SYM(CompileProgram); AIE_RTC_SYM(GetProgramLogSize); AIE_RTC_SYM(GetProgramLog);
  AIE_RTC_SYM(GetCodeSize); AIE_RTC_SYM(GetCode); AIE_RTC_SYM(DestroyProgram); AIE_RTC_SYM(Version);
#undef AIE_RTC_SYM
  Dl_info info;
  if (r.CreateProgram && dladdr(reinterpret_cast<void*>(r.CreateProgram), &info) && info.dli_fname) r.path = info.dli_fname;
  return r.CreateProgram && r.CompileProgram && r.GetCodeSize && r.GetCode && r.DestroyProgram;
}

static inline bool read_file(const std::string& path, std::string& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[65536];
  size_t n;
  out.clear();
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
  fclose(f);
  return true;
}
static inline uint64_t fnv1a(uint64_t h, const void* data, size_t n) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
  return h;
}
static inline std::string dirname_of(const std::string& p) {
  const size_t k = p.rfind('/');
  return k == std::string::npos ? std::string(".") : p.substr(0, k);
}
static inline bool is_dir(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
static inline void mkdirs(const std::string& p) {  // (private to the user: the files in it are code that gets loaded and run)
  for (size_t k = 1; k <= p.size(); ++k)
    if (k == p.size() || p[k] == '/') mkdir(p.substr(0, k).c_str(), 0700);
}
// a cache directory is only used if it belongs to this user and nobody else can write to it
static inline bool private_dir(const std::string& p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & (S_IWGRP | S_IWOTH)) == 0;
}
static const char* const kRecipe = "recipe 2: -O3 -std=c++17 -Wno-comment";  // compile options: part of the cache key

static const char* const kSources[] = {"aie_kernels.hip", "aie_kernels_ose.hip", "aie_layout.h", "aie_glibc_math.h",
                                       "aie_glibc_tables.h"};

// Compiles (or fetches from the cache) the code object for `image` (sizeof(aie_params) normalised bytes); `waves` =
// waves per SIMD the kernel is compiled for.  On failure returns false with a message in `err`.
// `ignore_cache`: compile even if a cached object exists (it failed to load) and replace it.
static inline bool code_object(const void* image, size_t image_bytes, int waves, const char* arch, bool ose,
                               std::string& code, std::string& err, bool* from_cache, bool ignore_cache = false) {
  // where the sources live: beside this library (in-tree build), or AIE_JIT_SOURCE_DIR
  std::string csrc;
  if (const char* e = getenv("AIE_JIT_SOURCE_DIR")) csrc = e;
  else {
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&fnv1a), &info) && info.dli_fname) csrc = dirname_of(info.dli_fname);
  }
  std::string inc = csrc + "/../../include", text;
  Rtc rtc;  // (loaded up front: its version is part of the key -- a code object of another toolchain is not reused)
  const bool have_rtc = load_rtc(rtc);
  int rtc_version[2] = {0, 0};
  if (have_rtc && rtc.Version) (void)rtc.Version(&rtc_version[0], &rtc_version[1]);
  uint64_t h = fnv1a(1469598103934665603ull, image, image_bytes);
  h = fnv1a(h, kRecipe, strlen(kRecipe));
  h = fnv1a(h, rtc_version, sizeof(rtc_version));
  const char* pad_nops = getenv("AIE_JIT_PAD_NOPS");  // development: code placement experiment (aie_kernels.hip)
  if (pad_nops) h = fnv1a(h, pad_nops, strlen(pad_nops));
  h = fnv1a(h, &waves, sizeof(waves));
  h = fnv1a(h, &ose, sizeof(ose));
  h = fnv1a(h, arch, strlen(arch));
  for (const char* s : kSources) {
    if (!read_file(csrc + "/" + s, text)) { err = "kernel source " + csrc + "/" + s + " not found"; return false; }
    h = fnv1a(h, text.data(), text.size());
  }
  if (!read_file(inc + "/aie.h", text)) { err = "include/aie.h not found beside the kernel sources"; return false; }
  h = fnv1a(h, text.data(), text.size());
  std::string cache;
  if (const char* e = getenv("AIE_JIT_CACHE")) cache = e;
  else if (const char* x = getenv("XDG_CACHE_HOME")) cache = std::string(x) + "/ai_economist_amd";
  else if (const char* home = getenv("HOME")) cache = std::string(home) + "/.cache/ai_economist_amd";
  else cache = "/tmp/ai_economist_amd_cache_" + std::to_string((long)geteuid());
  mkdirs(cache);
  const bool use_cache = private_dir(cache);  // (somebody else's or a world-writable directory: compile every time instead)
  char name[64];
  snprintf(name, sizeof(name), "/jit_%016llx.hsaco", (unsigned long long)h);
  const std::string file = cache + name;
  if (use_cache && ignore_cache) unlink(file.c_str());
  if (use_cache && !ignore_cache && read_file(file, code) && code.size() > 1024) {
    if (from_cache) *from_cache = true;
    return true;
  }
  if (from_cache) *from_cache = false;
  if (!have_rtc) { err = "libhiprtc.so could not be loaded"; return false; }
  // one compiler per code object and cache directory: the ranks of a multi-process launch all miss the cache at the same
  // moment; the others wait on the lock file and then find the object the first one wrote
  struct FileLock {
    int fd = -1;
    ~FileLock() { if (fd >= 0) { (void)flock(fd, LOCK_UN); close(fd); } }
  } lock;
  if (use_cache) {
    lock.fd = open((file + ".lock").c_str(), O_CREAT | O_RDWR, 0600);
    if (lock.fd >= 0 && flock(lock.fd, LOCK_EX) == 0 && !ignore_cache && read_file(file, code) && code.size() > 1024) {
      if (from_cache) *from_cache = true;
      return true;
    }
  }
  // the image as a header: exactly the shape of aie_spec_generated.h, one instance
  std::string hdr = "#pragma once\n#define AIE_N_SPECS 1\ntemplate <int K> struct aie_spec_image;\n"
                    "alignas(16) static constexpr unsigned char aie_jit_bytes[" + std::to_string(image_bytes) + "] = {";
  const unsigned char* ib = static_cast<const unsigned char*>(image);
  for (size_t i = 0; i < image_bytes; ++i) {
    hdr += std::to_string((int)ib[i]);
    hdr += (i % 32 == 31) ? ",\n" : ",";
  }
  hdr += "};\ntemplate <> struct aie_spec_image<0> { static constexpr const unsigned char* bytes = aie_jit_bytes; "
         "static constexpr int waves = " + std::to_string(waves) + "; };\n";
  const std::string src = ose ? "#define AIE_JIT 1\n#define AIE_JIT_OSE 1\n#include \"aie_kernels_ose.hip\"\n"
                              : "#define AIE_JIT 1\n#include \"aie_kernels.hip\"\n";
  const char* headers[1] = {hdr.c_str()};
  const char* header_names[1] = {"aie_jit_image.h"};
  hiprtcProgram prog = nullptr;
  if (rtc.CreateProgram(&prog, src.c_str(), "aie_jit.hip", 1, headers, header_names) != 0) { err = "hiprtcCreateProgram failed"; return false; }
  // include paths: our sources, the HIP headers and clang's own headers of the ROCm tree hiprtc came from, libc's
  std::vector<std::string> opts = {std::string("--offload-arch=") + arch, "-O3", "-std=c++17", "-Wno-comment",
                                   "-I" + csrc, "-I" + inc};
  if (pad_nops) opts.push_back(std::string("-DAIE_JIT_PAD_NOPS=") + pad_nops);
  std::string rocm = getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "";
  if (rocm.empty() && !rtc.path.empty()) rocm = dirname_of(dirname_of(rtc.path));
  if (rocm.empty() || !is_dir(rocm + "/include/hip")) rocm = "/opt/rocm";
  opts.push_back("-I" + rocm + "/include");
  const std::string clang_root = rocm + "/lib/llvm/lib/clang";
  if (DIR* d = opendir(clang_root.c_str())) {
    while (struct dirent* de = readdir(d))
      if (de->d_name[0] != '.' && is_dir(clang_root + "/" + de->d_name + "/include"))
        opts.push_back("-I" + clang_root + "/" + de->d_name + "/include");
    closedir(d);
  }
  for (const char* sys : {"/usr/include", "/usr/include/x86_64-linux-gnu"})
    if (is_dir(sys)) opts.push_back(std::string("-I") + sys);
  std::vector<const char*> optv;
  for (const std::string& o : opts) optv.push_back(o.c_str());
  const int rc = rtc.CompileProgram(prog, (int)optv.size(), optv.data());
  if (rc != 0) {
    size_t n = 0;
    std::string log;
    if (rtc.GetProgramLogSize && rtc.GetProgramLog && rtc.GetProgramLogSize(prog, &n) == 0 && n > 1) {
      log.resize(n);
      rtc.GetProgramLog(prog, &log[0]);
    }
    err = "hiprtc could not compile the specialised kernels: " + log.substr(0, 300);
    rtc.DestroyProgram(&prog);
    return false;
  }
  size_t n = 0;
  rtc.GetCodeSize(prog, &n);
  code.resize(n);
  rtc.GetCode(prog, &code[0]);
  rtc.DestroyProgram(&prog);
  // cache it (temporary name + rename: concurrent ranks compile the same thing and race harmlessly)
  if (!use_cache) return true;
  const std::string tmp = file + ".tmp" + std::to_string((long)getpid());
  if (FILE* f = fopen(tmp.c_str(), "wb")) {
    (void)chmod(tmp.c_str(), 0600);
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), file.c_str()) != 0) unlink(tmp.c_str());
  }
  return true;
}


// ---- background specialisation (aie_create starts one for every configuration without a compile-time instance) ----
// A job outlives its environment if need be (the thread holds a reference): destroying an environment never waits for
// a compiler.  At most two compile at a time; environments of the same configuration share one job.
struct Job {
  std::vector<unsigned char> image;
  int waves = 0;
  bool ose = false;
  std::string arch, code, err;
  std::atomic<int> state{0};  // 0 running, 1 code ready, -1 failed
  long pid = (long)getpid();  // the process whose thread compiles it (a forked child does not wait for the parent's jobs)
  std::atomic<int> claim{0};  // who compiles: 0 nobody yet (queued), 1 the background thread, 2 a caller that waits for
                              // the result anyway (aie_specialize: no point queueing behind other environments' jobs)
};
struct JobTable {
  std::mutex mu;
  std::condition_variable cv;
  int running = 0;
  bool shutting_down = false;  // process exit: queued jobs give up, the exit waits for the (at most two) running ones
  std::map<uint64_t, std::weak_ptr<Job>> jobs;
};
static inline JobTable*& job_table_slot() {
  static JobTable* t = nullptr;
  return t;
}
static inline JobTable& job_table() {
  static const bool once = [] {
    job_table_slot() = new JobTable();  // (never destroyed: detached threads may outlive static destructors)
    // A compiler thread should not be inside hiprtc when the process tears its libraries down: queued jobs give up at
    // once, a running compile gets $AIE_JIT_EXIT_WAIT_S seconds (default 10; a compile takes about 5) to finish.
    atexit([] {
      JobTable& T = job_table();
      const char* w = getenv("AIE_JIT_EXIT_WAIT_S");
      const int secs = w ? atoi(w) : 10;
      std::unique_lock<std::mutex> lock(T.mu);
      T.shutting_down = true;
      T.cv.notify_all();
      T.cv.wait_for(lock, std::chrono::seconds(secs > 0 ? secs : 0), [&T] { return T.running == 0; });
    });
    // fork(): the child has none of the parent's compiler threads, and the table's mutex may have been locked by one of
    // them at that moment -- the child starts with a fresh table (jobs of inherited environments are simply never
    // adopted there: they stay on the generic kernel unless the child calls aie_specialize)
    pthread_atfork(nullptr, nullptr, [] { job_table_slot() = new JobTable(); });
    return true;
  }();
  (void)once;
  return *job_table_slot();
}
static inline std::shared_ptr<Job> start_job(const void* image, size_t image_bytes, int waves, const char* arch, bool ose) {
  JobTable& T = job_table();
  uint64_t key = fnv1a(1469598103934665603ull, image, image_bytes);
  key = fnv1a(key, &waves, sizeof(waves));
  key = fnv1a(key, arch, strlen(arch));
  std::lock_guard<std::mutex> lock(T.mu);
  auto it = T.jobs.find(key);
  if (it != T.jobs.end())
    if (std::shared_ptr<Job> j = it->second.lock()) return j;
  std::shared_ptr<Job> job = std::make_shared<Job>();
  job->image.assign(static_cast<const unsigned char*>(image), static_cast<const unsigned char*>(image) + image_bytes);
  job->waves = waves;
  job->ose = ose;
  job->arch = arch;
  T.jobs[key] = job;
  std::thread([job]() {
    JobTable& T = job_table();
    {
      std::unique_lock<std::mutex> lock(T.mu);
      T.cv.wait(lock, [&T] { return T.running < 2 || T.shutting_down; });
      if (T.shutting_down) {
        job->err = "process exit";
        job->state.store(-1, std::memory_order_release);
        return;
      }
      int nobody = 0;
      if (!job->claim.compare_exchange_strong(nobody, 1)) return;  // a waiting caller took the job over
      T.running += 1;
    }
    bool cached = false;
    const bool ok = code_object(job->image.data(), job->image.size(), job->waves, job->arch.c_str(), job->ose, job->code,
                                job->err, &cached);
    {
      std::lock_guard<std::mutex> lock(T.mu);
      T.running -= 1;
    }
    T.cv.notify_all();
    job->state.store(ok ? 1 : -1, std::memory_order_release);
  }).detach();
  return job;
}

// A caller that waits for `job` anyway compiles it itself if the background thread has not started on it.
static inline void run_job_now_if_queued(const std::shared_ptr<Job>& job) {
  int nobody = 0;
  if (!job->claim.compare_exchange_strong(nobody, 2)) return;
  bool cached = false;
  const bool ok = code_object(job->image.data(), job->image.size(), job->waves, job->arch.c_str(), job->ose, job->code, job->err,
                              &cached);
  job->state.store(ok ? 1 : -1, std::memory_order_release);
}

}  // namespace aie_jit
