// host/mbank.hpp — llpf_mbank: a sweep of independent filters sharded over the GPUs of one node.  Part of capi.hip.
//
// Reference analogue: `map(svec) do s ... loglik(pfs,u,y) end` (test/runtests.jl:412-417) and metropolis_threaded, one
// filter per thread (src/smoothing.jl:335-347).  Filters never interact, so filter k lives on shard k mod n_shards and the
// path has exactly one exchange: the all-reduce (sum) of the per-filter log-likelihood vector, in which every shard fills
// only its own slots (== an all-gather), after a run.  That collective is RCCL over xGMI:
//   * one process, n_devices GPUs   — ncclCommInitAll + one grouped ncclAllReduce per run (llpf_mbank_create);
//   * one process per GPU (rank r of `world`) — ncclCommInitRank with an id the host distributes (llpf_mbank_unique_id on
//     rank 0, then llpf_mbank_create_rank everywhere): the layout torchrun / MPI / Julia Distributed give.
// librccl is loaded with dlopen on first use: a process that never builds a multi-GPU bank does not depend on it.
// Without a communicator (one shard; or create_rank with id == NULL, where the caller owns the exchange; or a device list
// that names one GPU twice — RCCL refuses that — where the shards' vectors are summed on the host) no RCCL call is made.
// A slot is written by exactly one shard and is zero elsewhere, so the sum is exact: sharded and unsharded sweeps give the
// same bits (tests/test_gpu_mbank.py).
#include <dlfcn.h>

#include <mutex>
#include <thread>

namespace rccl_dl {
typedef struct ncclComm* comm_t;
typedef struct { char internal[128]; } unique_id;
typedef int result_t;
enum { Success = 0, Float64 = 8, Sum = 0 };        // ncclDouble == ncclFloat64 == 8, ncclSum == 0 (rccl.h)
struct Api {
    void* handle = nullptr;
    result_t (*GetUniqueId)(unique_id*) = nullptr;
    result_t (*CommInitRank)(comm_t*, int, unique_id, int) = nullptr;
    result_t (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    result_t (*CommDestroy)(comm_t) = nullptr;
    result_t (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    result_t (*GroupStart)() = nullptr;
    result_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(result_t) = nullptr;
    std::string err, path;
};
static void load(Api& a);
static Api* api() {            // loaded once, also when two host threads build their first multi-GPU bank at the same time
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] { load(a); });
    return &a;
}
static void load(Api& a) {
    // RCCL must sit on the SAME HIP runtime as this library: streams and device pointers of one libamdhip64 mean nothing to
    // another, and a process can hold two of them (PyTorch wheels bundle their own next to their own librccl).  So: find the
    // file the runtime this library is bound to was loaded from, and take the librccl of that directory; only then the names.
    std::vector<std::string> names;
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            names.push_back(dir + "librccl.so.1");
            names.push_back(dir + "librccl.so");
        }
    }
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    for (const std::string& n : names) { a.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND); if (a.handle) { a.path = n; break; } }
    if (!a.handle) { a.err = std::string("librccl not loadable: ") + dlerror(); return; }
#define LLPF_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name)); if (!a.field) { a.err = std::string("librccl lacks ") + name; a.handle = nullptr; return; }
    LLPF_SYM(GetUniqueId, "ncclGetUniqueId") LLPF_SYM(CommInitRank, "ncclCommInitRank") LLPF_SYM(CommInitAll, "ncclCommInitAll")
    LLPF_SYM(CommDestroy, "ncclCommDestroy") LLPF_SYM(AllReduce, "ncclAllReduce") LLPF_SYM(GroupStart, "ncclGroupStart")
    LLPF_SYM(GroupEnd, "ncclGroupEnd") LLPF_SYM(GetErrorString, "ncclGetErrorString")
#undef LLPF_SYM
}
}  // namespace rccl_dl

#define RCCLC(expr)                                                                                              \
    do {                                                                                                         \
        rccl_dl::result_t _r = (expr);                                                                           \
        if (_r != rccl_dl::Success)                                                                              \
            return fail(LLPF_ERR_HIP, std::string(#expr) + ": " + rccl_dl::api()->GetErrorString(_r));           \
    } while (0)

struct MShard {
    Bank bank;
    int device = 0;
    std::vector<int> owned;          // global filter indices, ascending: shard s owns s, s + S, s + 2S, ...
    double* d_ll = nullptr;          // [n_filters] on the shard's device: this shard's slots filled, zero elsewhere
    rccl_dl::comm_t comm = nullptr;
    bool created = false;
};

enum { MBANK_COLL_NONE = 0, MBANK_COLL_RCCL = 1, MBANK_COLL_HOST = 2, MBANK_COLL_EXTERNAL = 3 };

struct llpf_mbank {
    int n_filters = 0;
    int n_shards_total = 1;          // shards of the sweep over all processes
    int first_shard = 0;             // global index of shards[0] (rank in the one-process-per-GPU layout)
    int collective = MBANK_COLL_NONE;
    std::vector<std::unique_ptr<MShard>> shards;     // the shards this process drives
    std::vector<double> h_ll;        // [n_filters] staging
    double last_run_ms = 0.0;        // slowest local shard
    double last_coll_ms = 0.0;       // host time of the collective (enqueue + completion)
};

// run fn(shard index) for every local shard, one host thread per shard when there are several (a handle is driven by one
// thread at a time; distinct handles are independent); the first failure's status and message are returned
template <class Fn>
static int mbank_foreach(llpf_mbank& m, Fn fn) {
    const int S = (int)m.shards.size();
    if (S == 1) return fn(0);
    std::vector<int> rc(S, LLPF_OK);
    std::vector<std::string> msg(S);
    std::vector<std::thread> th;
    th.reserve(S);
    // nothing may leave a thread's function as an exception (std::terminate), and a thread that cannot be started is a status:
    // the shards already started are joined, the others never run
    int spawn_rc = LLPF_OK;
    for (int s = 0; s < S && spawn_rc == LLPF_OK; ++s) {
        try {
            test_throw("thread");
            th.emplace_back([&, s]() noexcept {
                try { test_throw("shard"); rc[s] = fn(s); } catch (...) { rc[s] = guard_catch("shard worker"); }
                if (rc[s] != LLPF_OK) { try { msg[s] = g_err; } catch (...) {} }
            });
        } catch (...) {
            spawn_rc = guard_catch("starting a shard's host thread");
        }
    }
    for (auto& t : th) t.join();
    if (spawn_rc != LLPF_OK) return spawn_rc;
    for (int s = 0; s < S; ++s)
        if (rc[s] != LLPF_OK) return fail(rc[s], "shard " + std::to_string(m.first_shard + s) + " (device " + std::to_string(m.shards[s]->device) + "): " + msg[s]);
    return LLPF_OK;
}

static void mbank_free(llpf_mbank* m) {
    if (!m) return;
    for (auto& sp : m->shards) {
        MShard& sh = *sp;
        hipSetDevice(sh.device);
        if (sh.comm && rccl_dl::api()->CommDestroy) rccl_dl::api()->CommDestroy(sh.comm);
        if (sh.d_ll) hipFree(sh.d_ll);
        if (sh.created) free_bank(sh.bank);
    }
    delete m;
}

// shard `gs` of `S` owns the filters k with k mod S == gs
static void mbank_owned(int n_filters, int gs, int S, std::vector<int>& out) {
    out.clear();
    for (int k = gs; k < n_filters; k += S) out.push_back(k);
}

static int mbank_build(llpf_mbank* m, const llpf_config* base, const llpf_model* models, int n_filters,
                       const int32_t* devices, int n_local, int first_shard, int n_shards_total) {
    if (!base) return fail(LLPF_ERR_ARG, "null config");
    if (n_filters < n_shards_total) return fail(LLPF_ERR_ARG, "n_filters must be >= the number of shards (every shard needs a filter)");
    m->n_filters = n_filters;
    m->n_shards_total = n_shards_total;
    m->first_shard = first_shard;
    m->h_ll.assign((size_t)n_filters, 0.0);
    for (int s = 0; s < n_local; ++s) {
        m->shards.emplace_back(new MShard());
        MShard& sh = *m->shards.back();
        sh.device = devices[s];
        mbank_owned(n_filters, first_shard + s, n_shards_total, sh.owned);
    }
    CHK(mbank_foreach(*m, [&](int s) -> int {
        MShard& sh = *m->shards[s];
        llpf_config c = *base;
        c.device = sh.device;
        std::vector<llpf_model> mine;
        if (models) { mine.reserve(sh.owned.size()); for (int k : sh.owned) mine.push_back(models[k]); }
        sh.created = true;
        CHK(bank_create(&c, models ? mine.data() : nullptr, (int)sh.owned.size(), sh.bank, (uint64_t)(first_shard + s), (uint64_t)n_shards_total));
        HIPC(hipMalloc(&sh.d_ll, sizeof(double) * (size_t)n_filters));
        HIPC(hipMemset(sh.d_ll, 0, sizeof(double) * (size_t)n_filters));
        return LLPF_OK;
    }));
    return LLPF_OK;
}

// the exchange: on return h_ll holds every filter's log-likelihood (or, MBANK_COLL_EXTERNAL, this process's slots and zeros)
static int mbank_exchange(llpf_mbank& m, const std::vector<std::vector<double>>& local) {
    const size_t nb = sizeof(double) * (size_t)m.n_filters;
    const int S = (int)m.shards.size();
    if (m.collective != MBANK_COLL_RCCL) {
        // no communicator: one shard, shards sharing a GPU (host sum), or the caller reduces (external)
        std::fill(m.h_ll.begin(), m.h_ll.end(), 0.0);
        for (int s = 0; s < S; ++s)
            for (size_t i = 0; i < m.shards[s]->owned.size(); ++i) m.h_ll[(size_t)m.shards[s]->owned[i]] += local[s][i];
        return LLPF_OK;
    }
    std::vector<double> stage((size_t)m.n_filters);
    for (int s = 0; s < S; ++s) {
        MShard& sh = *m.shards[s];
        std::fill(stage.begin(), stage.end(), 0.0);
        for (size_t i = 0; i < sh.owned.size(); ++i) stage[(size_t)sh.owned[i]] = local[s][i];
        HIPC(hipSetDevice(sh.device));
        HIPC(hipMemcpyAsync(sh.d_ll, stage.data(), nb, hipMemcpyHostToDevice, sh.bank.stream));
        HIPC(hipStreamSynchronize(sh.bank.stream));      // `stage` is reused for the next shard
    }
    rccl_dl::Api* R = rccl_dl::api();
    (void)hipGetLastError();                              // RCCL reads the runtime's sticky last-error: start clean
    if (S > 1) RCCLC(R->GroupStart());
    for (int s = 0; s < S; ++s) {
        MShard& sh = *m.shards[s];
        HIPC(hipSetDevice(sh.device));
        RCCLC(R->AllReduce(sh.d_ll, sh.d_ll, (size_t)m.n_filters, rccl_dl::Float64, rccl_dl::Sum, sh.comm, sh.bank.stream));
    }
    if (S > 1) RCCLC(R->GroupEnd());
    MShard& s0 = *m.shards[0];
    HIPC(hipSetDevice(s0.device));
    HIPC(hipMemcpyAsync(m.h_ll.data(), s0.d_ll, nb, hipMemcpyDeviceToHost, s0.bank.stream));
    for (int s = 0; s < S; ++s) {
        HIPC(hipSetDevice(m.shards[s]->device));
        HIPC(hipStreamSynchronize(m.shards[s]->bank.stream));
    }
    return LLPF_OK;
}

static int mbank_run(llpf_mbank& m, const double* U, const double* Y, int64_t T, double t_index0, double* ll_total, double* ll_sum, bool aux, int aux_mode) {
    const int S = (int)m.shards.size();
    std::vector<std::vector<double>> local(S);
    for (int s = 0; s < S; ++s) local[s].assign(m.shards[s]->owned.size(), 0.0);
    CHK(mbank_foreach(m, [&](int s) -> int {
        Bank& b = m.shards[s]->bank;
        if (aux) return bank_aux_run(b, U, Y, T, aux_mode, local[s].data(), nullptr, nullptr, nullptr, nullptr, nullptr);
        return bank_run(b, U, Y, T, t_index0, local[s].data(), nullptr, nullptr, nullptr, nullptr, nullptr);
    }));
    m.last_run_ms = 0.0;
    for (int s = 0; s < S; ++s) m.last_run_ms = std::max(m.last_run_ms, m.shards[s]->bank.last_run_ms);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    CHK(mbank_exchange(m, local));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    m.last_coll_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    double sum = 0.0;
    for (int k = 0; k < m.n_filters; ++k) { if (ll_total) ll_total[k] = m.h_ll[(size_t)k]; sum = sum + m.h_ll[(size_t)k]; }   // fixed order
    if (ll_sum) *ll_sum = sum;
    return LLPF_OK;
}
