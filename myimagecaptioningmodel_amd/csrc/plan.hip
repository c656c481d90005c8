// Host side of libcapmi.so that is not a kernel launcher:
//   * capmi_plan_run: walks a packed launch table (entry id + argument slots + lane + record/wait marks) and enqueues a
//     whole train step -- ~650 launches on up to four HIP streams -- in ONE call from the host language, instead of one
//     foreign call per launch (the Python host spent 7-8 ms per 9.7 ms step in ctypes);
//   * capmi_comm_* / capmi_allreduce_bucket: the gradient all-reduce of ParallelExecutor
//     (/root/reference/ImageCaptioning/train.py:121-124) over RCCL, as an entry point that can sit in the same table.
#include <dlfcn.h>

#include <tuple>
#include <type_traits>
#include <utility>

#include "common.h"

// ------------------------------------------------------------------ argument slots -> typed call
namespace {

template <typename T> inline T from_slot(uint64_t s) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(s));
    } else if constexpr (std::is_same_v<T, float>) {
        uint32_t u = static_cast<uint32_t>(s);
        float f;
        memcpy(&f, &u, 4);
        return f;
    } else {
        return static_cast<T>(static_cast<int64_t>(s));      // int / int64_t / long long, two's complement in the slot
    }
}

template <typename F> struct Sig;
template <typename... A> struct Sig<int (*)(A...)> {
    static constexpr int arity = sizeof...(A);
    using Tup = std::tuple<A...>;
    // every entry point ends in `void* stream`: the slots hold the arguments before it
    template <int (*Fn)(A...), size_t... I> static int call(const uint64_t* s, void* stream, std::index_sequence<I...>) {
        return Fn(from_slot<std::tuple_element_t<I, Tup>>(s[I])..., stream);
    }
};

typedef int (*Thunk)(const uint64_t*, void*);
template <auto Fn> int thunk(const uint64_t* s, void* stream) {
    using S = Sig<decltype(Fn)>;
    return S::template call<Fn>(s, stream, std::make_index_sequence<S::arity - 1>{});
}

struct Entry {
    const char* name;
    Thunk fn;
    int nargs;      // without the stream
};
#define CAPMI_ENTRY(f) {#f, &thunk<&f>, Sig<decltype(&f)>::arity - 1}

const Entry g_entries[] = {
    CAPMI_ENTRY(capmi_igemm_nt),
    CAPMI_ENTRY(capmi_igemm_nt_group),
    CAPMI_ENTRY(capmi_igemm_nt_splitk),
    CAPMI_ENTRY(capmi_igemm_nt_bn),
    CAPMI_ENTRY(capmi_igemm_nt_bnact),
    CAPMI_ENTRY(capmi_igemm_nt_bnfin),
    CAPMI_ENTRY(capmi_igemm_nt_bnred),
    CAPMI_ENTRY(capmi_igemm_nt_bnsum),
    CAPMI_ENTRY(capmi_igemm_nt_stat),
    CAPMI_ENTRY(capmi_bn_stat_apply),
    CAPMI_ENTRY(capmi_bn_stat_apply_pool),
    CAPMI_ENTRY(capmi_igemm_tn_wgrad),
    CAPMI_ENTRY(capmi_colsum),
    CAPMI_ENTRY(capmi_im2col_stem),
    CAPMI_ENTRY(capmi_s2d_stem),
    CAPMI_ENTRY(capmi_s2d_stem_mask_grad),
    CAPMI_ENTRY(capmi_dwconv3x3_fwd),
    CAPMI_ENTRY(capmi_dwconv3x3_bwd_data),
    CAPMI_ENTRY(capmi_dwconv3x3_bwd_weight),
    CAPMI_ENTRY(capmi_maxpool3x3s2_fwd),
    CAPMI_ENTRY(capmi_maxpool3x3s2_bwd),
    CAPMI_ENTRY(capmi_bn_stats),
    CAPMI_ENTRY(capmi_bn_finalize),
    CAPMI_ENTRY(capmi_bn_apply),
    CAPMI_ENTRY(capmi_bn_apply_mask),
    CAPMI_ENTRY(capmi_bn_inference_coef),
    CAPMI_ENTRY(capmi_bn_inference_coef_batched),
    CAPMI_ENTRY(capmi_bn_finalize_apply),
    CAPMI_ENTRY(capmi_bn_bwd_reduce),
    CAPMI_ENTRY(capmi_bn_bwd_reduce_final),
    CAPMI_ENTRY(capmi_bn_bwd_apply),
    CAPMI_ENTRY(capmi_bn_bwd_reduce_spread),
    CAPMI_ENTRY(capmi_bn_bwd_apply_spread),
    CAPMI_ENTRY(capmi_bn_bwd_reduce_pool),
    CAPMI_ENTRY(capmi_bn_bwd_apply_pool),
    CAPMI_ENTRY(capmi_bn_bwd_reduce_pool_x),
    CAPMI_ENTRY(capmi_bn_bwd_apply_pool_x),
    CAPMI_ENTRY(capmi_add_act),
    CAPMI_ENTRY(capmi_act_bwd),
    CAPMI_ENTRY(capmi_mean_rows),
    CAPMI_ENTRY(capmi_mean_rows_bwd),
    CAPMI_ENTRY(capmi_embedding_fwd),
    CAPMI_ENTRY(capmi_caption_feed),
    CAPMI_ENTRY(capmi_embedding_bwd),
    CAPMI_ENTRY(capmi_bcast_rows),
    CAPMI_ENTRY(capmi_bcast_rows_bwd),
    CAPMI_ENTRY(capmi_lstm_cell_fwd),
    CAPMI_ENTRY(capmi_decode_prep),
    CAPMI_ENTRY(capmi_lstm_cell_sentinel_fwd),
    CAPMI_ENTRY(capmi_lstm_cell_bwd),
    CAPMI_ENTRY(capmi_lstm_step_fwd),
    CAPMI_ENTRY(capmi_lstm_step_bwd),
    CAPMI_ENTRY(capmi_lstm_seq_fwd),
    CAPMI_ENTRY(capmi_lstm_seq_bwd),
    CAPMI_ENTRY(capmi_sentinel_fwd),
    CAPMI_ENTRY(capmi_sentinel_bwd),
    CAPMI_ENTRY(capmi_ada_attention_fwd),
    CAPMI_ENTRY(capmi_ada_attention_bwd),
    CAPMI_ENTRY(capmi_softmax_xent_fwd),
    CAPMI_ENTRY(capmi_xent_finalize),
    CAPMI_ENTRY(capmi_softmax_xent_bwd),
    CAPMI_ENTRY(capmi_argmax),
    CAPMI_ENTRY(capmi_beam_step),
    CAPMI_ENTRY(capmi_gather_rows),
    CAPMI_ENTRY(capmi_beam_backtrack),
    CAPMI_ENTRY(capmi_adam),
    CAPMI_ENTRY(capmi_adam_shadow),
    CAPMI_ENTRY(capmi_adam_g16),
    CAPMI_ENTRY(capmi_cast),
    CAPMI_ENTRY(capmi_weight_dgrad_form),
    CAPMI_ENTRY(capmi_weight_dgrad_form_batched),
    CAPMI_ENTRY(capmi_fill_f32),
    CAPMI_ENTRY(capmi_allreduce_bucket),
    CAPMI_ENTRY(capmi_allreduce_bucket_bf16),
};
constexpr int kEntries = sizeof(g_entries) / sizeof(g_entries[0]);

}      // namespace

extern "C" int capmi_plan_entry_count(void) { return kEntries; }
extern "C" const char* capmi_plan_entry_name(int i) { return (i >= 0 && i < kEntries) ? g_entries[i].name : nullptr; }
extern "C" int capmi_plan_entry_nargs(int i) { return (i >= 0 && i < kEntries) ? g_entries[i].nargs : -1; }

extern "C" int capmi_plan_run(const capmi_launch* table, int n, void* const* streams, int nstreams) {
    CAPMI_CHECK(table && streams && n >= 0 && nstreams >= 1, "capmi_plan_run: bad arguments");
    for (int i = 0; i < n; ++i) {
        const capmi_launch& L = table[i];
        CAPMI_CHECK(L.lane >= 0 && L.lane < nstreams, "capmi_plan_run: row %d: lane %d of %d", i, L.lane, nstreams);
        void* stream = streams[L.lane];
        if (L.kind == CAPMI_PLAN_LAUNCH) {
            CAPMI_CHECK(L.entry >= 0 && L.entry < kEntries, "capmi_plan_run: row %d: entry %d", i, L.entry);
            const Entry& e = g_entries[L.entry];
            CAPMI_CHECK(L.nargs == e.nargs, "capmi_plan_run: row %d: %s takes %d arguments, table has %d", i, e.name, e.nargs, L.nargs);
            int rc = e.fn(L.args, stream);
            if (rc != 0) return rc;      // the entry point has set the error text
        } else if (L.kind == CAPMI_PLAN_RECORD) {
            hipError_t err = hipEventRecord((hipEvent_t)(uintptr_t)L.args[0], (hipStream_t)stream);
            CAPMI_CHECK(err == hipSuccess, "capmi_plan_run: row %d: record: %s", i, hipGetErrorString(err));
        } else if (L.kind == CAPMI_PLAN_WAIT) {
            hipError_t err = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)(uintptr_t)L.args[0], 0);
            CAPMI_CHECK(err == hipSuccess, "capmi_plan_run: row %d: wait: %s", i, hipGetErrorString(err));
        } else {
            CAPMI_CHECK(false, "capmi_plan_run: row %d: kind %d", i, L.kind);
        }
    }
    return 0;
}

// ------------------------------------------------------------------ RCCL (ncclAllReduce over xGMI)
// The process already holds ONE librccl (the one torch.distributed's "nccl" backend uses); a second copy in the same
// address space would bring its own topology state and its own proxy threads.  So the library is looked up at run
// time among the objects already mapped, and only loaded by name when none is.
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.handle) return 0;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names)
        if ((h = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char* nm : names)
            if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    CAPMI_CHECK(h, "capmi_comm: librccl.so is not loadable: %s", dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(h, "ncclCommUserRank");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    CAPMI_CHECK(g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.GetErrorString,
                "capmi_comm: librccl.so lacks an expected symbol");
    g_rccl.handle = h;
    return 0;
}
}      // namespace

static_assert(sizeof(ncclUniqueId) == CAPMI_COMM_ID_BYTES, "capmi.h: CAPMI_COMM_ID_BYTES != sizeof(ncclUniqueId)");

extern "C" int capmi_comm_unique_id(void* id_out) {
    CAPMI_CHECK(id_out, "capmi_comm_unique_id: null pointer");
    if (rccl_load()) return 1;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    CAPMI_CHECK(r == ncclSuccess, "capmi_comm_unique_id: %s", g_rccl.GetErrorString(r));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

extern "C" int capmi_comm_init(void** comm, int nranks, int rank, const void* id) {
    CAPMI_CHECK(comm && id && nranks >= 1 && rank >= 0 && rank < nranks, "capmi_comm_init: bad arguments");
    if (rccl_load()) return 1;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&c, nranks, uid, rank);
    CAPMI_CHECK(r == ncclSuccess, "capmi_comm_init: %s", g_rccl.GetErrorString(r));
    *comm = (void*)c;
    return 0;
}

/* What RCCL itself says about the communicator: ranks in it and this process's rank (ncclCommCount / ncclCommUserRank). */
extern "C" int capmi_comm_count(void* comm, int* nranks, int* rank) {
    CAPMI_CHECK(comm && nranks && rank, "capmi_comm_count: null pointer");
    CAPMI_CHECK(g_rccl.CommCount && g_rccl.CommUserRank, "capmi_comm_count: librccl.so lacks ncclCommCount / ncclCommUserRank");
    ncclResult_t r = g_rccl.CommCount((ncclComm_t)comm, nranks);
    CAPMI_CHECK(r == ncclSuccess, "capmi_comm_count: %s", g_rccl.GetErrorString(r));
    r = g_rccl.CommUserRank((ncclComm_t)comm, rank);
    CAPMI_CHECK(r == ncclSuccess, "capmi_comm_count: %s", g_rccl.GetErrorString(r));
    return 0;
}

extern "C" int capmi_comm_destroy(void* comm) {
    if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)comm);
    return 0;
}

extern "C" int capmi_allreduce_bucket(void* comm, float* buf, int64_t n, void* stream) {
    CAPMI_CHECK(comm && g_rccl.AllReduce, "capmi_allreduce_bucket: no communicator (capmi_comm_init first)");
    if (n <= 0) return 0;
    CAPMI_CHECK(buf, "capmi_allreduce_bucket: null buffer");
    ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    CAPMI_CHECK(r == ncclSuccess, "capmi_allreduce_bucket: %s", g_rccl.GetErrorString(r));
    return 0;
}

extern "C" int capmi_allreduce_bucket_bf16(void* comm, void* buf16, int64_t n, void* stream) {
    CAPMI_CHECK(comm && g_rccl.AllReduce, "capmi_allreduce_bucket_bf16: no communicator (capmi_comm_init first)");
    if (n <= 0) return 0;
    CAPMI_CHECK(buf16, "capmi_allreduce_bucket_bf16: null buffer");
    ncclResult_t r = g_rccl.AllReduce(buf16, buf16, (size_t)n, ncclBfloat16, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    CAPMI_CHECK(r == ncclSuccess, "capmi_allreduce_bucket_bf16: %s", g_rccl.GetErrorString(r));
    return 0;
}
