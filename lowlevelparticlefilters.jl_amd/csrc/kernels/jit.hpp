// kernels/jit.hpp — user-supplied models, compiled at run time (host side; part of kernels.hip, namespace llpf).
// ------------------------------------------------------------------------------------------------
// The reference's filters take arbitrary `dynamics` / `measurement` callables (src/PFtypes.jl:59-63, 189-193); a Julia
// closure cannot run on the GPU, but a device-code snippet can: llpf_model_compile() takes HIP source that defines
//     struct UserModel {                                   // the Model concept of kernels/models.hpp
//         static constexpr bool RB = false;
//         DEV void prepare(const ModelD* m, const double* u, double t);   // particle-independent terms; m->A, B, C, qt[16], Ts,
//                                                                          // supersample are the model's parameter block
//         DEV void dynamics(const double* x, double* out) const;          // f(x, u, p, t) without noise
//         DEV void measurement(const double* x, double* out) const;       // g(x, u, p, t)
//     };
// appends it to the engine's own headers (jit_prelude.inc, generated at build time) and compiles k_step<UserModel, nx, ny, MODE>
// with hiprtc (--offload-arch of the device, -ffp-contract=off like the engine itself).  Filters with such a model run the
// balanced form: the precompiled k_resample + the compiled k_step; noise and likelihood stay the Gaussian descriptors.
// ------------------------------------------------------------------------------------------------
// (kernels.hip includes <hip/hiprtc.h>, <mutex>, <vector> and jit_prelude.inc before it opens the namespace)

struct JitModel {
    int nx = 0, ny = 0;
    std::string src;                           // the snippet: a second llpf_model_compile of the same (source, nx, ny) returns the same id
    std::vector<char> code;
    static constexpr int NK = 12;
    std::string name[NK];                      // lowered names of k_step<UserModel, nx, ny, MODE, STEP_PPT>, MODE = 0..3; [4]: k_user_bound<UserModel>;
                                               // [5]: k_smooth_fx<UserModel, nx, ny>; models whose dynamics are worth a table (marks): [6], [7]:
                                               // k_step<..., MODE_PROP / MODE_PROP_WEIGHT, STEP_PPT, true>; [8], [9]: k_resample_fx<UserModel, nx, systematic / stratified>;
                                               // [10]: k_init_user<UserModel, nx>; [11]: k_traits_tag<model_traits<UserModel>::value> (never launched)
    bool marks = false;
    int traits = 0;                            // LLPF_TRAIT_*: the optional members the snippet defines
    struct PerDevice { hipModule_t mod = nullptr; hipFunction_t fn[NK] = {}; };
    std::vector<PerDevice> dev;                // indexed by device ordinal, loaded on first use
};
static std::mutex g_jit_mutex;
static std::vector<JitModel*> g_jit_models;

static int jit_compile_model(const char* device_src, int nx, int ny, bool internal, std::string& err);
int jit_compile_user_model(const char* device_src, int nx, int ny, std::string& err) { return jit_compile_model(device_src, nx, ny, false, err); }
// The linear-Gaussian model at dimensions the library was not precompiled for (nx or ny in 5..16; the reference is generic in the
// state dimension, src/PFtypes.jl:65-75): the engine's own LinGauss<NX, NY> compiled on demand like a user model, cached per shape.
// Such filters run the balanced form (k_resample + the compiled k_step); k_init / k_norm / the smoother's draw exist for 1..16, the auxiliary filter's second half for 1..8.
int jit_builtin_lg(int nx, int ny, std::string& err) {
    const std::string src = "struct UserModel : LinGauss<" + std::to_string(nx) + ", " + std::to_string(ny) + "> {};\n"
                            "template <> struct share_dynamics<UserModel> { static constexpr bool value = false; };\n";
    return jit_compile_model(src.c_str(), nx, ny, true, err);
}
static int jit_compile_model(const char* device_src, int nx, int ny, bool internal, std::string& err) {
    if (!device_src) { err = "null source"; return -1; }
    // the kernels around the compiled k_step (k_init, k_norm with the weighted mean, k_resample, the auxiliary second half) are
    // precompiled for 1..4 state and measurement dimensions
    if (!internal && (nx < 1 || nx > 4 || ny < 1 || ny > 4)) { err = "user models: nx and ny must be in 1..4"; return -1; }
    if (nx < 1 || nx > MAXD || ny < 1 || ny > MAXD) { err = "nx, ny must be in 1..16"; return -1; }
    {   // parameter sweeps and PMMH loops rebuild filters with the same snippet: compile once
        std::lock_guard<std::mutex> lk(g_jit_mutex);
        for (size_t k = 0; k < g_jit_models.size(); ++k)
            if (g_jit_models[k]->nx == nx && g_jit_models[k]->ny == ny && g_jit_models[k]->src == device_src) return LLPF_MODEL_USER_BASE + (int)k;
    }
    std::string src(LLPF_JIT_PRELUDE);
    src += "\nnamespace llpf {\n";
    src += device_src;
    src += "\n}  // namespace llpf\n";
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "llpf_user_model.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { err = "hiprtcCreateProgram failed"; return -1; }
    std::string expr[JitModel::NK];
    for (int mode = 0; mode < 4; ++mode)
        expr[mode] = "llpf::k_step<llpf::UserModel, " + std::to_string(nx) + ", " + std::to_string(ny) + ", " + std::to_string(mode) + ", " + std::to_string(STEP_PPT) + ">";
    expr[4] = "llpf::k_user_bound<llpf::UserModel>";
    expr[5] = "llpf::k_smooth_fx<llpf::UserModel, " + std::to_string(nx) + ", " + std::to_string(ny) + ">";
    // a user's dynamics are taken to be worth a table (share_dynamics, kernels/models.hpp): the resampling launch evaluates them once per
    // surviving source and the step kernel gathers (kernels/resfx.hpp)
    // (not the internally generated linear-Gaussian snippet, nx or ny above 4: kernels/models.hpp, marks_path)
    const bool marks = !internal;
    const int nk = marks ? JitModel::NK : 6;
    if (marks) {
        expr[10] = "llpf::k_init_user<llpf::UserModel, " + std::to_string(nx) + ">";
        expr[11] = "llpf::k_traits_tag<llpf::model_traits<llpf::UserModel>::value>";
        for (int m = 0; m < 2; ++m)
            expr[6 + m] = "llpf::k_step<llpf::UserModel, " + std::to_string(nx) + ", " + std::to_string(ny) + ", " + std::to_string(m == 0 ? (int)MODE_PROP : (int)MODE_PROP_WEIGHT) + ", " + std::to_string(STEP_PPT) + ", true>";
        expr[8] = "llpf::k_resample_fx<llpf::UserModel, " + std::to_string(nx) + ", " + std::to_string((int)LLPF_RESAMPLE_SYSTEMATIC) + ">";
        expr[9] = "llpf::k_resample_fx<llpf::UserModel, " + std::to_string(nx) + ", " + std::to_string((int)LLPF_RESAMPLE_STRATIFIED) + ">";
    }
    for (int mode = 0; mode < nk; ++mode) hiprtcAddNameExpression(prog, expr[mode].c_str());
    int devid = 0;
    hipDeviceProp_t prop;
    std::string arch = "gfx950";
    if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess && prop.gcnArchName[0]) arch = prop.gcnArchName;
    const std::string archopt = "--offload-arch=" + arch;
    // -disable-machine-licm: as for k_step.hip (Makefile) — the tile loop of k_step<..., MARKS>
    // (-DLLPF_EXP_LDEXP: as for k_step.hip in the Makefile — the same bits, one instruction instead of six for exp's scaling)
    const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-DLLPF_EXP_LDEXP=1", "-mllvm", "-disable-machine-licm"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        err = std::string("hiprtc: ") + hiprtcGetErrorString(rc) + "\n" + log;
        hiprtcDestroyProgram(&prog);
        return -1;
    }
    JitModel* jm = new JitModel();
    jm->nx = nx; jm->ny = ny; jm->src = device_src; jm->marks = marks;
    size_t sz = 0;
    hiprtcGetCodeSize(prog, &sz);
    jm->code.resize(sz);
    hiprtcGetCode(prog, jm->code.data());
    for (int mode = 0; mode < nk; ++mode) {
        const char* low = nullptr;
        if (hiprtcGetLoweredName(prog, expr[mode].c_str(), &low) != HIPRTC_SUCCESS || !low) { err = "hiprtcGetLoweredName failed for " + expr[mode]; delete jm; hiprtcDestroyProgram(&prog); return -1; }
        jm->name[mode] = low;
    }
    if (marks) {      // "...k_traits_tagILi<value>EEEvv": what the snippet provides, without running anything
        const size_t at = jm->name[11].find("k_traits_tagILi");
        if (at == std::string::npos) { err = "could not read the model's traits from " + jm->name[11]; delete jm; hiprtcDestroyProgram(&prog); return -1; }
        jm->traits = atoi(jm->name[11].c_str() + at + 15);
        if (jm->traits & LLPF_TRAIT_NOISE) jm->marks = false;      // a model that forms its own noise needs x next to f(x): inline dynamics
    }
    hiprtcDestroyProgram(&prog);
    std::lock_guard<std::mutex> lk(g_jit_mutex);
    // two threads may have compiled the same snippet side by side: the first registration wins, "the same (source, nx, ny) returns the same id"
    for (size_t k = 0; k < g_jit_models.size(); ++k)
        if (g_jit_models[k]->nx == nx && g_jit_models[k]->ny == ny && g_jit_models[k]->src == device_src) { delete jm; return LLPF_MODEL_USER_BASE + (int)k; }
    g_jit_models.push_back(jm);
    return LLPF_MODEL_USER_BASE + (int)g_jit_models.size() - 1;
}

static JitModel* jit_model(int model_id) {
    std::lock_guard<std::mutex> lk(g_jit_mutex);
    const int k = model_id - LLPF_MODEL_USER_BASE;
    return (k >= 0 && k < (int)g_jit_models.size()) ? g_jit_models[(size_t)k] : nullptr;
}
static bool jit_supported(int model_id, int nx, int ny) {
    JitModel* jm = jit_model(model_id);
    return jm && jm->nx == nx && jm->ny == ny;
}
static hipError_t jit_function(JitModel* jm, int which, hipFunction_t* fn) {      // this device's handle of kernel `which`
    int devid = 0;
    hipError_t e = hipGetDevice(&devid);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_jit_mutex);
    if ((int)jm->dev.size() <= devid) jm->dev.resize((size_t)devid + 1);
    JitModel::PerDevice& pd = jm->dev[(size_t)devid];
    if (!pd.mod && (e = hipModuleLoadData(&pd.mod, jm->code.data())) != hipSuccess) return e;
    if (!pd.fn[which] && (e = hipModuleGetFunction(&pd.fn[which], pd.mod, jm->name[which].c_str())) != hipSuccess) return e;
    *fn = pd.fn[which];
    return hipSuccess;
}
// a user model with a likelihood of its own: its upper bound into every filter's descriptor (see k_user_bound); a no-op kernel otherwise
hipError_t launch_user_bound(int model_id, ModelD* models, int F, const double* zero_u, hipStream_t s) {
    JitModel* jm = jit_model(model_id);
    if (!jm) return hipErrorInvalidValue;
    hipFunction_t fn = nullptr;
    hipError_t e = jit_function(jm, 4, &fn);
    if (e != hipSuccess) return e;
    void* args[] = {&models, &zero_u};
    return hipModuleLaunchKernel(fn, (unsigned)F, 1, 1, 64, 1, 1, 0, s, args, nullptr);
}
hipError_t launch_smooth_fx_user(const BankDev& b, const SmoothArgs& a, hipStream_t s) {
    JitModel* jm = jit_model(b.model_id);
    if (!jm) return hipErrorInvalidValue;
    hipFunction_t fn = nullptr;
    hipError_t e = jit_function(jm, 5, &fn);
    if (e != hipSuccess) return e;
    BankDev bd = b;
    const ModelD* models = b.models;
    SmoothArgs aa = a;
    void* args[] = {&bd, &models, &aa};
    return hipModuleLaunchKernel(fn, (unsigned)((b.N + BLOCK - 1) / BLOCK), 1, 1, BLOCK, 1, 1, 0, s, args, nullptr);
}
int jit_model_traits(int model_id) { JitModel* jm = jit_model(model_id); return jm ? jm->traits : -1; }
// reset! of a model with an initial density of its own (UserModel::initial)
hipError_t launch_init_user(const BankDev& b, const double* zero_u, uint32_t step, int init_anc, hipStream_t s) {
    JitModel* jm = jit_model(b.model_id);
    if (!jm || !(jm->traits & LLPF_TRAIT_INITIAL)) return hipErrorInvalidValue;
    hipFunction_t fn = nullptr;
    hipError_t e = jit_function(jm, 10, &fn);
    if (e != hipSuccess) return e;
    BankDev bd = b;
    const ModelD* models = b.models;
    const FilterScal* scal = b.scal;
    void* args[] = {&bd, &models, &scal, &zero_u, &step, &init_anc};
    return hipModuleLaunchKernel(fn, (unsigned)((b.Ns + BLOCK - 1) / BLOCK), (unsigned)b.F, 1, BLOCK, 1, 1, 0, s, args, nullptr);
}
static unsigned step_grid_x_marks(const BankDev& b);      // k_step.hip
static bool jit_marks(int model_id) { JitModel* jm = jit_model(model_id); return jm && jm->marks; }
static hipError_t launch_step_user(const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    JitModel* jm = jit_model(b.model_id);
    if (!jm || mode < 0 || mode > 3) return hipErrorInvalidValue;
    const bool marks = a.marks && jm->marks && (mode == MODE_PROP || mode == MODE_PROP_WEIGHT);      // the form that follows k_resample_fx
    if (marks && (!b.mark || !b.fxs)) return hipErrorInvalidValue;
    hipFunction_t fn = nullptr;
    hipError_t e = jit_function(jm, marks ? (mode == MODE_PROP ? 6 : 7) : mode, &fn);
    if (e != hipSuccess) return e;
    BankDev bd = b;
    const ModelD* models = b.models;
    const FilterScal* scal = b.scal;
    StepArgs aa = a;
    void* args[] = {&bd, &models, &scal, &aa};
    const unsigned gx = marks ? step_grid_x_marks(b) : (unsigned)(b.Ns / (BLOCK * STEP_PPT));
    return hipModuleLaunchKernel(fn, gx, (unsigned)b.F, 1, BLOCK, 1, 1, 0, s, args, nullptr);
}
static hipError_t launch_resample_fx_user(const BankDev& b, const ResArgs& a, const StepArgs& st, hipStream_t s) {
    JitModel* jm = jit_model(b.model_id);
    if (!jm || !jm->marks) return hipErrorInvalidValue;
    hipFunction_t fn = nullptr;
    hipError_t e = jit_function(jm, b.strategy == LLPF_RESAMPLE_SYSTEMATIC ? 8 : 9, &fn);
    if (e != hipSuccess) return e;
    BankDev bd = b;
    ResArgs aa = a;
    StepArgs ss = st;
    void* args[] = {&bd, &aa, &ss};
    return hipModuleLaunchKernel(fn, (unsigned)b.P2, (unsigned)b.F, 1, BLOCK, 1, 1, 0, s, args, nullptr);
}

// ---- k_rbfull for shapes the library was not precompiled for (kernels/rbfull.hpp is part of the prelude) -----------------------------
struct JitRbfull {
    int fk = 0, nn = 0, nl = 0, ny = 0;
    std::vector<char> code;
    std::string name[3];                       // MODE_WEIGHT, MODE_PROP, MODE_PROP_WEIGHT
    struct PerDevice { hipModule_t mod = nullptr; hipFunction_t fn[3] = {nullptr, nullptr, nullptr}; };
    std::vector<PerDevice> dev;
};
static std::vector<JitRbfull*> g_jit_rbfull;
static JitRbfull* jit_rbfull_find(int fk, int nn, int nl, int ny) {
    for (JitRbfull* j : g_jit_rbfull) if (j->fk == fk && j->nn == nn && j->nl == nl && j->ny == ny) return j;
    return nullptr;
}
int jit_prepare_rbfull(int fk, int nn, int nl, int ny, std::string& err) {
    {
        std::lock_guard<std::mutex> lk(g_jit_mutex);
        if (jit_rbfull_find(fk, nn, nl, ny)) return 0;
    }
    std::string src(LLPF_JIT_PRELUDE);
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "llpf_rbfull_shape.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { err = "hiprtcCreateProgram failed"; return -1; }
    const std::string model = (fk == 1 ? "llpf::QuadTank<" : "llpf::LinGauss<") + std::to_string(nn) + ", " + std::to_string(ny) + ">";
    std::string expr[3];
    for (int mode = 0; mode < 3; ++mode) {
        expr[mode] = "llpf::k_rbfull<" + model + ", " + std::to_string(nn) + ", " + std::to_string(nl) + ", " + std::to_string(ny) + ", " + std::to_string(mode) + ">";
        hiprtcAddNameExpression(prog, expr[mode].c_str());
    }
    int devid = 0;
    hipDeviceProp_t prop;
    std::string arch = "gfx950";
    if (hipGetDevice(&devid) == hipSuccess && hipGetDeviceProperties(&prop, devid) == hipSuccess && prop.gcnArchName[0]) arch = prop.gcnArchName;
    const std::string archopt = "--offload-arch=" + arch;
    // -disable-machine-licm: as for k_rbfull.hip (Makefile) — hoisted out of the persistent loop, the literal constants of the body are spilled
    const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-value", "-mllvm", "-disable-machine-licm"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, (int)(sizeof(opts) / sizeof(opts[0])), opts);
    if (rc != HIPRTC_SUCCESS) {
        size_t n = 0;
        hiprtcGetProgramLogSize(prog, &n);
        std::string log(n, '\0');
        if (n) hiprtcGetProgramLog(prog, &log[0]);
        err = std::string("hiprtc: ") + hiprtcGetErrorString(rc) + "\n" + log;
        hiprtcDestroyProgram(&prog);
        return -1;
    }
    JitRbfull* j = new JitRbfull();
    j->fk = fk; j->nn = nn; j->nl = nl; j->ny = ny;
    size_t sz = 0;
    hiprtcGetCodeSize(prog, &sz);
    j->code.resize(sz);
    hiprtcGetCode(prog, j->code.data());
    for (int mode = 0; mode < 3; ++mode) {
        const char* low = nullptr;
        if (hiprtcGetLoweredName(prog, expr[mode].c_str(), &low) != HIPRTC_SUCCESS || !low) { err = "hiprtcGetLoweredName failed for " + expr[mode]; delete j; hiprtcDestroyProgram(&prog); return -1; }
        j->name[mode] = low;
    }
    hiprtcDestroyProgram(&prog);
    std::lock_guard<std::mutex> lk(g_jit_mutex);
    if (jit_rbfull_find(fk, nn, nl, ny)) { delete j; return 0; }      // another thread registered the shape meanwhile
    g_jit_rbfull.push_back(j);
    return 0;
}
hipError_t launch_rbfull_jit(int fk, int nn, int nl, int ny, const BankDev& b, int mode, const StepArgs& a, hipStream_t s) {
    if (mode < 0 || mode > 2) return hipErrorInvalidValue;
    int devid = 0;
    hipError_t e = hipGetDevice(&devid);
    if (e != hipSuccess) return e;
    hipFunction_t fn = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_jit_mutex);
        JitRbfull* j = jit_rbfull_find(fk, nn, nl, ny);
        if (!j) return hipErrorInvalidValue;            // jit_prepare_rbfull runs when the bank is built
        if ((int)j->dev.size() <= devid) j->dev.resize((size_t)devid + 1);
        JitRbfull::PerDevice& pd = j->dev[(size_t)devid];
        if (!pd.mod && (e = hipModuleLoadData(&pd.mod, j->code.data())) != hipSuccess) return e;
        if (!pd.fn[mode] && (e = hipModuleGetFunction(&pd.fn[mode], pd.mod, j->name[mode].c_str())) != hipSuccess) return e;
        fn = pd.fn[mode];
    }
    BankDev bd = b;
    StepArgs aa = a;
    // LLPF_RBF_HOT_PARAMS (kernels/rbfull.hpp), then the two structs
    void* args[] = {&bd.scal, &bd.bank_flag, &bd.anc, &bd.models, &aa.u, &bd.Ns, &bd.nu, &aa.u_stride, &aa.y, &bd, &aa};
    return hipModuleLaunchKernel(fn, rbfull_grid_x(b, nl, mode), (unsigned)b.F, 1, 64, 1, 1, 0, s, args, nullptr);
}
