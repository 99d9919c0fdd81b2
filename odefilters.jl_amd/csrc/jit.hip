// Run-time compiled vector fields.
// The reference calls a Julia closure f (and f.jac / ForwardDiff) in the middle of every step
// (src/perform_step.jl:106,116-121).  A device kernel cannot call back into the host, so a user vector field is
// handed over as SOURCE: a struct with the interface of the compiled-in registry (csrc/rhs.h) -- f generic in the
// scalar type, so that the same text serves the step (double) and the Taylor-mode initialisation (truncated jets,
// src/state_initialization.jl:2-53), plus the analytic Jacobian for EK1 (optional).  hipcc compiles the very same lane
// functions (ek_lane.h, smooth_lane.h, dense_lane.h, sample_lane.h, smooth_rows.h) around it for gfx950; the
// kernels are loaded with the module API and launched with the same parameter structs as the compiled-in ones.
#include <elf.h>
#include <hip/hip_runtime.h>

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>

#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "jit.h"

extern char** environ;

namespace odef {
namespace {

struct JitRhs {
  std::string name, source, include_dir;
  int d, np;
};

std::mutex g_mu;
std::vector<JitRhs> g_rhs;                                                      // id = kJitFirstId + index
std::map<std::tuple<int, int, int, int>, std::unique_ptr<JitModule>> g_modules;  // (id, q, ek1, device)
std::map<std::tuple<int, int, int>, const TeamLaunch*> g_teams;                  // (id, q, ek1): shared objects, never unloaded

// path of this library (the run-time compiled shared objects of the workgroup-per-trajectory path link against it)
std::string this_library() {
  Dl_info info;
  if (dladdr((const void*)&this_library, &info) && info.dli_fname) return info.dli_fname;
  return std::string();
}

// Where the kernel headers live: $ODEFILTER_HIP_INCLUDE, else `../csrc` next to the directory this library was loaded
// from (odefilters.jl_amd/lib/libodefilter_hip.so -> odefilters.jl_amd/csrc; found with dladdr, so a tree that was moved
// or copied after the build still compiles user vector fields), else the directory of the build.
std::string default_include_dir() {
  if (const char* e = getenv("ODEFILTER_HIP_INCLUDE")) return e;
  Dl_info info;
  if (dladdr((const void*)&default_include_dir, &info) && info.dli_fname) {
    std::string lib = info.dli_fname;
    const size_t slash = lib.rfind('/');
    const std::string dir = (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/../csrc";
    if (access((dir + "/ek_lane.h").c_str(), R_OK) == 0) return dir;
  }
#ifdef ODEF_DEFAULT_CSRC
  return ODEF_DEFAULT_CSRC;
#else
  return ".";
#endif
}

std::string translation_unit(const JitRhs& r, int q, int ek1, bool with_posterior_kernels) {
  const int D = r.d * (q + 1);
  std::string s;
  if (D > 12) s += "#define ODEF_ROWSTORE_FREE_OFFSET 1\n";  // see RowStore (ek_lane.h)
  s += "#include \"ek_lane.h\"\n";
  if (with_posterior_kernels) s += "#include \"smooth_lane.h\"\n#include \"dense_lane.h\"\n#include \"sample_lane.h\"\n";
  const bool rows_smoother = !with_posterior_kernels && D <= 32;  // 12 < D <= 32: the row-per-lane team smoother
  if (rows_smoother) s += "#include \"smooth_rows.h\"\n#include \"dense_rows.h\"\n#include \"sample_rows.h\"\n";
  const bool rows16 = D <= 16;  // the 16-lanes-per-trajectory filter and smoother of small ensembles (rows_kernels.h)
  if (rows16) s += "#include \"rows_kernels.h\"\n";
  s += "namespace odef {\n";
  s += r.source;
  s += "\nusing RhsJit = " + r.name + ";\n";
  s += "static_assert(RhsJit::d == " + std::to_string(r.d) + ", \"d of the struct differs from the d passed to odef_rhs_compile\");\n";
  s += "static_assert(RhsJit::np == " + std::to_string(r.np) + ", \"np of the struct differs from the n_params passed to odef_rhs_compile\");\n";
  const std::string Q = std::to_string(q), EK = ek1 ? "true" : "false", DD = std::to_string(r.d);
  s += "extern \"C\" __global__ __launch_bounds__(64) void odef_jit_fixed_every(const FilterParams P) {\n"
       "  const long i0 = (long)blockIdx.x * 64;\n"
       "  if (i0 + threadIdx.x < P.N) filter_fixed_lane<RhsJit, " + Q + ", " + EK + ", true>(P, i0, threadIdx.x);\n}\n";
  s += "extern \"C\" __global__ __launch_bounds__(64) void odef_jit_fixed_final(const FilterParams P) {\n"
       "  const long i0 = (long)blockIdx.x * 64;\n"
       "  if (i0 + threadIdx.x < P.N) filter_fixed_lane<RhsJit, " + Q + ", " + EK + ", false>(P, i0, threadIdx.x);\n}\n";
  s += "extern \"C\" __global__ __launch_bounds__(64) void odef_jit_adaptive(const FilterParams P) {\n"
       "  const long i0 = (long)blockIdx.x * 64;\n"
       "  if (i0 + threadIdx.x < P.N) filter_adaptive_lane<RhsJit, " + Q + ", " + EK + ">(P, i0, threadIdx.x);\n}\n";
  if (with_posterior_kernels) {
    const std::string TRI = std::to_string(D * (D + 1) / 2);
    for (int adapt = 0; adapt < 2; ++adapt) {
      s += std::string("extern \"C\" __global__ __launch_bounds__(64) void odef_jit_smooth_") + (adapt ? "adapt" : "fixed") +
           "(const SmoothParams P) {\n"
           "  __shared__ double lds[" + TRI + " * 64];\n"
           "  const long i0 = (long)blockIdx.x * 64;\n"
           "  const LaneMem xl{lds + threadIdx.x, 64};\n"
           "  const bool valid = i0 + threadIdx.x < P.N;\n"
           "  long n_hi = P.n_save;\n" +
           (adapt ? "  n_hi = wave_uniform_max(valid ? (long)P.nsaved[i0 + threadIdx.x] : 0, valid);\n" : "") +
           "  if (valid) smooth_lane_v2<" + DD + ", " + Q + ", " + (adapt ? "true" : "false") + ">(P, i0, threadIdx.x, xl, n_hi);\n}\n";
    }
    s += "extern \"C\" __global__ __launch_bounds__(64) void odef_jit_dense(const DenseParams P) {\n"
         "  __shared__ double lds[" + TRI + " * 64];\n"
         "  const long i = (long)blockIdx.x * 64 + threadIdx.x;\n"
         "  const LaneMem xl{lds + threadIdx.x, 64};\n"
         "  if (i < P.N) dense_lane<" + DD + ", " + Q + ">(P, i, (long)blockIdx.y, xl);\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(64) void odef_jit_sample(const SampleParams P) {\n"
         "  __shared__ double lds[" + TRI + " * 64];\n"
         "  const long i = (long)blockIdx.x * 64 + threadIdx.x;\n"
         "  const LaneMem xl{lds + threadIdx.x, 64};\n"
         "  const bool valid = i < P.N;\n"
         "  const long n_hi = (P.adaptive && !P.tq) ? wave_uniform_max(valid ? (long)P.nsaved[i] : 0, valid) : P.n_save;\n"
         "  if (valid) sample_lane<" + DD + ", " + Q + ">(P, i, (long)blockIdx.y, xl, n_hi);\n}\n";
  }
  if (rows_smoother) {
    const std::string TEAM = D <= 16 ? "16" : "32";
    s += "extern \"C\" __global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(128))) void odef_jit_smooth_rows(const SmoothParams P) {\n"
         "  constexpr int TEAM = " + TEAM + ", TPB = 64 / TEAM;\n"
         "  using W = RowsWs<" + DD + ", " + Q + " + 1>;\n"
         "  __shared__ double lds[TPB * W::size];\n"
         "  const int team = threadIdx.x / TEAM, tid = threadIdx.x % TEAM;\n"
         "  const long i = (long)blockIdx.x * TPB + team;\n"
         "  RowState<" + DD + " * (" + Q + " + 1)> st;\n"
         "  if (i < P.N) smooth_rows_lane<" + DD + ", " + Q + ", TEAM>(P, i, tid, lds + team * W::size, &st);\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(128))) void odef_jit_dense_rows(const DenseParams P) {\n"
         "  constexpr int TEAM = " + TEAM + ", TPB = 64 / TEAM;\n"
         "  using W = RowsWs<" + DD + ", " + Q + " + 1>;\n"
         "  __shared__ double lds[TPB * W::size];\n"
         "  const int team = threadIdx.x / TEAM, tid = threadIdx.x % TEAM;\n"
         "  const long it = (long)blockIdx.x * TPB + team;\n"
         "  RowState<" + DD + " * (" + Q + " + 1)> st;\n"
         "  if (it < P.N * P.n_q) dense_rows_lane<" + DD + ", " + Q + ", TEAM>(P, it % P.N, it / P.N, tid, lds + team * W::size, &st);\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(128))) void odef_jit_sample_rows(const SampleParams P) {\n"
         "  constexpr int TEAM = " + TEAM + ", TPB = 64 / TEAM;\n"
         "  using W = RowsWs<" + DD + ", " + Q + " + 1>;\n"
         "  __shared__ double lds[TPB * W::size];\n"
         "  const int team = threadIdx.x / TEAM, tid = threadIdx.x % TEAM;\n"
         "  const long it = (long)blockIdx.x * TPB + team;\n"
         "  RowState<" + DD + " * (" + Q + " + 1)> st;\n"
         "  if (it < P.N * P.n_samples) sample_rows_lane<" + DD + ", " + Q + ", TEAM>(P, it % P.N, it / P.N, tid, lds + team * W::size, &st);\n}\n";
  }
  if (rows16) {
    s += "extern \"C\" __global__ __launch_bounds__(256) void odef_jit_rows_fixed_every(const FilterParams P) {\n"
         "  rows_filter_fixed_entry<RhsJit, " + Q + ", " + EK + ", true>(P);\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void odef_jit_rows_fixed_final(const FilterParams P) {\n"
         "  rows_filter_fixed_entry<RhsJit, " + Q + ", " + EK + ", false>(P);\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void odef_jit_rows_adaptive(const FilterParams P) {\n"
         "  rows_filter_adaptive_entry<RhsJit, " + Q + ", " + EK + ">(P);\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void odef_jit_bcast_fixed(const SmoothParams P) {\n"
         "  rows_smooth_entry<" + DD + ", " + Q + ", false>(P);\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) void odef_jit_bcast_adapt(const SmoothParams P) {\n"
         "  rows_smooth_entry<" + DD + ", " + Q + ", true>(P);\n}\n";
  }
  s += "}  // namespace odef\n";
  return s;
}

// The workgroup-per-trajectory path (state dimension above 20, even d <= 32): the matrix-core filter (fixed grids and
// adaptive), the smoother (persistent and split pass), dense output and sampling of filter_mfma.h / smooth_mfma.h /
// dense_mfma.h / sample_mfma.h around the user's field, for ONE order and ONE algorithm, with their host-side launch code
// (team_launch_impl.h) -- exactly what inst_lorenz96.hip is for a compiled-in field.  The module exports the function table
// the C-ABI layer launches through.
std::string team_translation_unit(const JitRhs& r, int q, int ek1) {
  const std::string Q = std::to_string(q), EK = ek1 ? "true" : "false", DD = std::to_string(r.d);
  std::string s = "#include \"team_launch_impl.h\"\nnamespace odef {\n";
  s += r.source;
  s += "\nstruct RhsJit : " + r.name + " { static constexpr const char* name = \"" + r.name + "\"; };\n";
  s += "static_assert(RhsJit::d == " + DD + ", \"d of the struct differs from the d passed to odef_rhs_compile\");\n";
  s += "static_assert(RhsJit::np == " + std::to_string(r.np) + ", \"np of the struct differs from the n_params passed to odef_rhs_compile\");\n";
  s += "static int jit_filter_order(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive) {\n"
       "  if (q != " + Q + " || (ek1 != 0) != " + EK + ") return -2;\n"
       "  LaunchTilesFilterT<false> f{P, s, adaptive};\n"
       "  f.template operator()<RhsJit, " + Q + ", " + EK + ">();\n"
       "  return 0;\n}\n";
  s += "static int jit_filter(int q, int ek1, const FilterParams& P, hipStream_t s, int adaptive, double* stage, size_t stage_doubles, long* staged_recs) {\n"
       "  return team_filter_staged<" + DD + ">(q, ek1, P, s, adaptive, stage, stage_doubles, false, jit_filter_order, staged_recs);\n}\n";
  s += "static int jit_smooth(int q, const SmoothParams& P, double* ws, hipStream_t s) { return team_smooth_inplace<" + DD + ", " + Q + ">(q, P, ws, s); }\n";
  s += "static int jit_smooth_staged(int q, const SmoothParams& P, long n_rec, double* ws, double* stage, size_t stage_doubles, hipStream_t s, long in_stage) {\n"
       "  return team_smooth_staged<" + DD + ", " + Q + ">(q, P, n_rec, ws, stage, stage_doubles, s, in_stage);\n}\n";
  s += "static int jit_dense(int q, const DenseParams& P, double* ws, hipStream_t s) { return team_dense<" + DD + ", " + Q + ">(q, P, ws, s); }\n";
  s += "static int jit_sample(int q, const SampleParams& P, double* ws, hipStream_t s) { return team_sample<" + DD + ", " + Q + ">(q, P, ws, s); }\n";
  s += "static size_t jit_smooth_ws(int q) { return team_smooth_ws<" + DD + ", " + Q + ">(q); }\n";
  s += "}  // namespace odef\n";
  s += "extern \"C\" unsigned long odef_jit_abi() { return odef::team_abi_stamp(); }\n";
  s += "extern \"C\" const odef::TeamLaunch* odef_jit_team() {\n"
       "  using namespace odef;\n"
       "  static const TeamLaunch t = {" + DD + ", jit_filter, jit_smooth, jit_smooth_staged, jit_dense, jit_sample, jit_smooth_ws};\n"
       "  return &t;\n}\n";
  return s;
}

// What odef_rhs_compile builds for d > 10 (no lane kernel exists to try the text on): the vector field in double, in
// forward mode (the Jacobian EK1 needs when the struct has none) and on Taylor jets (the initialisation)
std::string probe_translation_unit(const JitRhs& r) {
  const std::string DD = std::to_string(r.d);
  std::string s = "#include \"ek_lane.h\"\nnamespace odef {\n";
  s += r.source;
  s += "\nusing RhsJit = " + r.name + ";\n";
  s += "static_assert(RhsJit::d == " + DD + ", \"d of the struct differs from the d passed to odef_rhs_compile\");\n";
  s += "static_assert(RhsJit::np == " + std::to_string(r.np) + ", \"np of the struct differs from the n_params passed to odef_rhs_compile\");\n";
  s += "extern \"C\" __global__ void odef_jit_probe(const double* u, const double* p, double* out) {\n"
       "  constexpr int d = " + DD + ";\n"
       "  double uu[d], du[d], J[d][d], m0[2 * d];\n"
       "  for (int a = 0; a < d; ++a) uu[a] = u[a];\n"
       "  RhsJit::f(uu, p, du);\n"
       "  rhs_jacobian<RhsJit>(uu, p, J);\n"
       "  taylor_init<RhsJit, 1>(uu, p, m0);\n"
       "  double acc = 0.0;\n"
       "  for (int a = 0; a < d; ++a) acc += du[a] + J[a][a] + m0[d + a];\n"
       "  out[0] = acc;\n}\n";
  s += "}  // namespace odef\n";
  return s;
}

// Source -> gfx950 code object; on failure `err` holds the compiler log.
// The compiler runs as a CHILD PROCESS (hipcc --genco), not in-process through hiprtc: hiprtc/comgr of ROCm 7.2 aborts
// the whole host process ("LLVM ERROR: Unsupported instruction") on the larger lane kernels (state dimension 14 and
// up), which the offline compiler builds without complaint; a child process can only fail with a log.
// $ODEFILTER_HIP_HIPCC overrides the compiler path (default: hipcc on PATH, then /opt/rocm/bin/hipcc).
std::string read_file(const std::string& path) {
  std::string out;
  if (FILE* f = fopen(path.c_str(), "rb")) {
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
  }
  return out;
}

// Names of the device functions a code object keeps OUT OF LINE (anything of type FUNC that is not one of our kernels).
// Such a function is compiled once, for the loosest register budget among its callers; called from a kernel with a
// tighter one (odef_jit_smooth_rows is pinned to 128 registers) it would address registers its wavefront does not own --
// a memory access fault at run time.  The generated kernels force-inline everything, and the build is refused otherwise.
std::string out_of_line_device_functions(const std::string& co) {
  const size_t at = co.find("\x7f" "ELF");
  if (at == std::string::npos || co.size() - at < sizeof(Elf64_Ehdr)) return "";
  const char* base = co.data() + at;
  const size_t avail = co.size() - at;
  Elf64_Ehdr eh;
  std::memcpy(&eh, base, sizeof eh);
  if (eh.e_shentsize != sizeof(Elf64_Shdr) || eh.e_shoff + (size_t)eh.e_shnum * sizeof(Elf64_Shdr) > avail) return "";
  std::string names;
  for (unsigned i = 0; i < eh.e_shnum; ++i) {
    Elf64_Shdr sh;
    std::memcpy(&sh, base + eh.e_shoff + (size_t)i * sizeof sh, sizeof sh);
    if (sh.sh_type != SHT_SYMTAB || sh.sh_link >= eh.e_shnum || sh.sh_offset + sh.sh_size > avail) continue;
    Elf64_Shdr st;
    std::memcpy(&st, base + eh.e_shoff + (size_t)sh.sh_link * sizeof st, sizeof st);
    if (st.sh_offset + st.sh_size > avail) continue;
    for (size_t k = 0; k + sizeof(Elf64_Sym) <= sh.sh_size; k += sizeof(Elf64_Sym)) {
      Elf64_Sym sym;
      std::memcpy(&sym, base + sh.sh_offset + k, sizeof sym);
      if (ELF64_ST_TYPE(sym.st_info) != STT_FUNC || sym.st_shndx == SHN_UNDEF || sym.st_name >= st.sh_size) continue;
      const char* nm = base + st.sh_offset + sym.st_name;
      const size_t len = strnlen(nm, st.sh_size - sym.st_name);
      if (len >= 9 && std::strncmp(nm, "odef_jit_", 9) == 0) continue;
      names += std::string(nm, len) + "\n";
    }
  }
  return names;
}

// The lane smoother (smooth_lane.h) keeps a packed matrix and the carried mean in a HAND-MANAGED file at the top of the
// lane's AGPRs, with register numbers fixed in inline assembly the compiler knows nothing about.  That is sound only while
// the compiler's own AGPR use (it parks VGPRs there under pressure, lowest register first) stays below the file.  The
// compiled-in kernels are checked at build time (tests/test_build_hygiene.py); a user's vector field changes nothing in
// that kernel, but the check costs nothing, so the run-time compiled ones are checked on their ISA listing here:
// returns the offending line, or "" when every AGPR reference of the odef_jit_smooth_* kernels outside the file's own
// assembly lies below `first_slot`.
std::string agpr_file_violation(const std::string& isa, int first_slot) {
  bool in_kernel = false, in_asm = false;
  size_t pos = 0;
  while (pos < isa.size()) {
    size_t eol = isa.find('\n', pos);
    if (eol == std::string::npos) eol = isa.size();
    const std::string line = isa.substr(pos, eol - pos);
    pos = eol + 1;
    if (line.rfind("odef_jit_smooth_", 0) == 0 && line.find(':') != std::string::npos) in_kernel = true;
    else if (line.rfind(".Lfunc_end", 0) == 0) in_kernel = false;
    if (!in_kernel) continue;
    if (line.find("ASMSTART") != std::string::npos) { in_asm = true; continue; }
    if (line.find("ASMEND") != std::string::npos) { in_asm = false; continue; }
    if (in_asm) continue;
    const std::string code = line.substr(0, line.find(';'));
    for (size_t k = 0; k + 1 < code.size(); ++k) {
      if (code[k] != 'a') continue;
      if (k > 0 && (isalnum((unsigned char)code[k - 1]) || code[k - 1] == '_')) continue;
      size_t j = k + 1;
      if (code[j] == '[') ++j;
      if (j >= code.size() || !isdigit((unsigned char)code[j])) continue;
      // every number of a register or register range: a12, a[12], a[12:13], a[0xdc]
      while (j < code.size() && (isalnum((unsigned char)code[j]) || code[j] == ':')) {
        if (isdigit((unsigned char)code[j])) {
          char* end = nullptr;
          const long r = strtol(code.c_str() + j, &end, 0);
          if (r >= first_slot) return line;
          j = (size_t)(end - code.c_str());
        } else {
          ++j;
        }
      }
    }
  }
  return "";
}

void remove_tree(const std::string& dir) {  // the compiler's temporaries (flat directory)
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string n = e->d_name;
      if (n != "." && n != "..") remove((dir + "/" + n).c_str());
    }
    closedir(d);
  }
  rmdir(dir.c_str());
}

// `agpr_first_slot` >= 0: the translation unit holds the lane smoother, whose AGPR file starts at that register (see above)
// shared_out != nullptr: build a host + device SHARED OBJECT instead of a code object (the workgroup-per-trajectory path: its
// host-side launch code, team_launch_impl.h, is compiled around the user's field too); *shared_out receives a handle from dlopen
bool compile(const std::string& src, const std::string& include_dir, std::vector<char>& code, std::string& err, int agpr_first_slot = -1,
             void** shared_out = nullptr) {
  char tmpl[] = "/tmp/odef_jit_XXXXXX";
  const char* dir = mkdtemp(tmpl);
  if (!dir) {
    err = "odef_rhs_compile: cannot create a temporary directory under /tmp";
    return false;
  }
  const std::string base = dir, srcp = base + "/rhs.hip", outp = base + (shared_out ? "/rhs.so" : "/rhs.co"), logp = base + "/log.txt";
  // (handed to the linker with -Wl: hipcc would take a bare path for another HIP source)
  const std::string self_path = shared_out ? this_library() : std::string(), self = "-Wl," + self_path;
  if (shared_out && self_path.empty()) {
    err = "odef_rhs_compile: cannot locate libodefilter_hip.so (dladdr) to link the run-time compiled module against";
    remove_tree(base);
    return false;
  }
  {
    FILE* f = fopen(srcp.c_str(), "wb");
    const bool written = f && fwrite(src.data(), 1, src.size(), f) == src.size();
    if (f && fclose(f) != 0) { /* reported below through `written` of the next open */ }
    if (!written) {
      err = "odef_rhs_compile: cannot write the generated source to " + srcp;
      remove_tree(base);
      return false;
    }
  }
  const std::string inc = "-I" + (include_dir.empty() ? default_include_dir() : include_dir);
  const char* env_cc = getenv("ODEFILTER_HIP_HIPCC");
  const char* candidates[] = {env_cc ? env_cc : "hipcc", "/opt/rocm/bin/hipcc"};
  int status = -1;
  bool spawned = false;
  for (const char* cc : candidates) {
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, logp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    // (--save-temps=obj: the ISA listing lands next to the output, for agpr_file_violation)
    const char* argv_co[] = {cc, "--offload-arch=gfx950", "-O3", "-std=c++20", "--genco", "-fno-crash-diagnostics", agpr_first_slot >= 0 ? "--save-temps=obj" : "-DODEF_NO_LISTING",
                             inc.c_str(), srcp.c_str(), "-o", outp.c_str(), nullptr};
    // (-amdgpu-function-calls=false: every device function inline -- kernels with different register budgets must not
    // share an out-of-line callee, see out_of_line_device_functions)
    const char* argv_so[] = {cc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-shared", "-fno-crash-diagnostics", "-mllvm", "-amdgpu-function-calls=false",
                             inc.c_str(), srcp.c_str(), "-o", outp.c_str(), self.c_str(), nullptr};
    // $ODEFILTER_HIP_JIT_FLAGS: extra compiler flags, space-separated (diagnostic builds: e.g. the LDS poisoning of filter_mfma.h)
    std::vector<std::string> extra;
    if (const char* ef = getenv("ODEFILTER_HIP_JIT_FLAGS")) {
      std::string tok;
      for (const char* c = ef;; ++c) {
        if (*c == ' ' || *c == 0) {
          if (!tok.empty()) extra.push_back(tok);
          tok.clear();
          if (*c == 0) break;
        } else
          tok += *c;
      }
    }
    std::vector<const char*> argv;
    for (const char* const* a = shared_out ? argv_so : argv_co; *a; ++a) {
      argv.push_back(*a);
      if (a == (shared_out ? argv_so : argv_co))  // right behind the compiler's name
        for (const auto& x : extra) argv.push_back(x.c_str());
    }
    argv.push_back(nullptr);
    pid_t pid = 0;
    const int rc = posix_spawnp(&pid, cc, &fa, nullptr, const_cast<char* const*>(argv.data()), environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) continue;
    spawned = true;
    while (waitpid(pid, &status, 0) < 0 && errno == EINTR) {
    }
    if (WIFEXITED(status) && WEXITSTATUS(status) == 127) {  // exec failed inside the child: try the next candidate
      spawned = false;
      continue;
    }
    break;
  }
  bool ok = spawned && WIFEXITED(status) && WEXITSTATUS(status) == 0;
  if (ok && shared_out) {
    // (the mapping outlives the file: the temporary directory goes away below)
    *shared_out = dlopen(outp.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!*shared_out) {
      const char* de = dlerror();
      err = std::string("odef_rhs_compile: dlopen of the run-time compiled module failed: ") + (de ? de : "?");
      remove_tree(base);
      return false;
    }
    remove_tree(base);
    return true;
  }
  if (ok) {
    const std::string co = read_file(outp);
    ok = !co.empty();
    code.assign(co.begin(), co.end());
    const std::string stray = ok ? out_of_line_device_functions(co) : std::string();
    if (!stray.empty()) {
      err = "odef_rhs_compile: the compiler left device functions out of line (kernels with different register budgets would share them):\n" + stray.substr(0, 4000);
      remove_tree(base);
      return false;
    }
    if (ok && agpr_first_slot >= 0) {
      const std::string isa = read_file(base + "/rhs-hip-amdgcn-amd-amdhsa-gfx950.s");
      const std::string bad = isa.empty() ? std::string("(no ISA listing was produced)") : agpr_file_violation(isa, agpr_first_slot);
      if (!bad.empty()) {
        err = "odef_rhs_compile: the compiler's register allocation reaches the smoother's hand-managed AGPR file (first slot a" +
              std::to_string(agpr_first_slot) + "): " + bad;
        remove_tree(base);
        return false;
      }
    }
  }
  if (!ok) {
    if (!spawned) {
      err = "odef_rhs_compile: cannot start hipcc (set ODEFILTER_HIP_HIPCC)";
    } else {
      const std::string log = read_file(logp);
      err = "hipcc: compilation of the user vector field failed\n";
      if (log.find("illegal VGPR to SGPR copy") != std::string::npos || log.find("ran out of registers") != std::string::npos)
        err += "(the lane-per-trajectory kernels keep the whole filter state of a trajectory in registers; this state dimension does not fit)\n";
      err += log.substr(0, 6000);
    }
  }
  remove_tree(base);
  return ok;
}

}  // namespace

int jit_register(const char* name, const char* source, int d, int np, const char* include_dir, std::string& err) {
  if (!name || !source || !*name) {
    err = "odef_rhs_compile: null or empty name/source";
    return -1;
  }
  if (d < 1 || d > 32 || np < 0) {
    err = "odef_rhs_compile: d must be in 1..32 (lane kernels: d(q+1) <= 20; above that the workgroup-per-trajectory kernels, even d) and n_params >= 0";
    return -1;
  }
  JitRhs r{name, source, include_dir ? include_dir : "", d, np};
  // compile the order-1 filter (d <= 10; a probe kernel above) once now so that errors in the user's text surface here, with the compiler log
  std::vector<char> code;
  if (!compile(d <= 10 ? translation_unit(r, 1, 1, false) : probe_translation_unit(r), r.include_dir, code, err)) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  g_rhs.push_back(std::move(r));
  return kJitFirstId + (int)g_rhs.size() - 1;
}

bool jit_lookup(int rhs_id, int* d, int* np) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int k = rhs_id - kJitFirstId;
  if (k < 0 || k >= (int)g_rhs.size()) return false;
  *d = g_rhs[k].d;
  *np = g_rhs[k].np;
  return true;
}

JitModule* jit_get_module(int rhs_id, int q, int ek1, int device, std::string& err) {
  // The hipcc child process takes seconds to minutes: it runs OUTSIDE the registry lock, so that odef_create for other
  // vector fields (jit_lookup, cached modules) is not blocked meanwhile.  Two threads asking for the same uncached
  // module may both compile; the first to publish wins, the other's module is unloaded.
  const auto key = std::make_tuple(rhs_id, q, ek1, device);
  JitRhs r;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    const int k = rhs_id - kJitFirstId;
    if (k < 0 || k >= (int)g_rhs.size()) {
      err = "unknown run-time rhs id";
      return nullptr;
    }
    auto it = g_modules.find(key);
    if (it != g_modules.end()) return it->second.get();
    r = g_rhs[k];  // copy: the vector may grow while we compile
  }
  auto m = std::make_unique<JitModule>();
  m->posterior = r.d * (q + 1) <= 12;  // the lane smoother / dense output / sampler keep a packed matrix per lane in LDS
  m->rows_team = r.d * (q + 1) <= 16 ? 16 : 32;
  m->rows16 = r.d * (q + 1) <= 16;
  std::vector<char> code;
  const int D = r.d * (q + 1);
  const int agpr_first_slot = m->posterior ? 2 * (128 - (D * (D + 1) / 2 + D)) : -1;  // MS of smooth_lane_v2 (smooth_lane.h)
  if (!compile(translation_unit(r, q, ek1, m->posterior), r.include_dir, code, err, agpr_first_slot)) return nullptr;
  const hipError_t le = hipModuleLoadData(&m->mod, code.data());
  if (le != hipSuccess) {
    err = std::string("hipModuleLoadData failed for the run-time compiled vector field: ") + hipGetErrorString(le);
    return nullptr;
  }
  struct { hipFunction_t* f; const char* name; bool need; } fn[] = {
      {&m->fixed_every, "odef_jit_fixed_every", true},   {&m->fixed_final, "odef_jit_fixed_final", true},
      {&m->adaptive, "odef_jit_adaptive", true},         {&m->smooth_fixed, "odef_jit_smooth_fixed", m->posterior},
      {&m->smooth_adapt, "odef_jit_smooth_adapt", m->posterior}, {&m->dense, "odef_jit_dense", m->posterior},
      {&m->sample, "odef_jit_sample", m->posterior},
      {&m->smooth_rows, "odef_jit_smooth_rows", !m->posterior && r.d * (q + 1) <= 32},
      {&m->dense_rows, "odef_jit_dense_rows", !m->posterior && r.d * (q + 1) <= 32},
      {&m->sample_rows, "odef_jit_sample_rows", !m->posterior && r.d * (q + 1) <= 32},
      {&m->rows_fixed_every, "odef_jit_rows_fixed_every", m->rows16},
      {&m->rows_fixed_final, "odef_jit_rows_fixed_final", m->rows16},
      {&m->rows_adaptive, "odef_jit_rows_adaptive", m->rows16},
      {&m->bcast_fixed, "odef_jit_bcast_fixed", m->rows16},
      {&m->bcast_adapt, "odef_jit_bcast_adapt", m->rows16}};
  for (auto& e : fn) {
    if (!e.need) continue;
    if (hipModuleGetFunction(e.f, m->mod, e.name) != hipSuccess) {
      err = std::string("kernel ") + e.name + " missing from the run-time compiled module";
      (void)hipModuleUnload(m->mod);
      return nullptr;
    }
  }
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_modules.find(key);
  if (it != g_modules.end()) {  // somebody else published it meanwhile
    (void)hipModuleUnload(m->mod);
    return it->second.get();
  }
  JitModule* out = m.get();
  g_modules[key] = std::move(m);
  return out;
}

const TeamLaunch* jit_get_team(int rhs_id, int q, int ek1, unsigned long abi_stamp, std::string& err) {
  // as jit_get_module: the compiler (minutes for these kernels) runs outside the registry lock
  const auto key = std::make_tuple(rhs_id, q, ek1);
  JitRhs r;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    const int k = rhs_id - kJitFirstId;
    if (k < 0 || k >= (int)g_rhs.size()) {
      err = "unknown run-time rhs id";
      return nullptr;
    }
    auto it = g_teams.find(key);
    if (it != g_teams.end()) return it->second;
    r = g_rhs[k];
  }
  std::vector<char> unused;
  void* handle = nullptr;
  if (!compile(team_translation_unit(r, q, ek1), r.include_dir, unused, err, -1, &handle)) return nullptr;
  using Entry = const TeamLaunch* (*)();
  Entry entry = (Entry)dlsym(handle, "odef_jit_team");
  if (!entry) {
    err = "odef_jit_team missing from the run-time compiled module";
    return nullptr;
  }
  using Stamp = unsigned long (*)();
  Stamp stamp = (Stamp)dlsym(handle, "odef_jit_abi");
  if (!stamp || stamp() != abi_stamp) {
    err = "the run-time compiled module was built from headers that do not match this library (parameter struct layouts differ): rebuild libodefilter_hip.so or point ODEFILTER_HIP_INCLUDE at its csrc";
    return nullptr;
  }
  const TeamLaunch* t = entry();
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_teams.find(key);
  if (it != g_teams.end()) return it->second;  // (somebody else published it meanwhile; the duplicate module stays loaded, unused)
  g_teams[key] = t;
  return t;
}

int jit_launch(hipFunction_t f, unsigned gx, unsigned gy, const void* params, hipStream_t s, unsigned block) {
  void* args[] = {const_cast<void*>(params)};
  return hipModuleLaunchKernel(f, gx, gy, 1, block, 1, 1, 0, s, args, nullptr) == hipSuccess ? 0 : -4;
}

}  // namespace odef
