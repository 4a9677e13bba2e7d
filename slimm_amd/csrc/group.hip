// Several GPUs in ONE process: a group of contexts, one per device, that behaves like one (include/slimm_hip.h,
// slimm_group_*).  The C++ host of SURVEY.md section 8e: records are dealt to the members BY READ while they are
// pushed, every member runs the phases on its own reads, and the two exchanges of the multi-rank path
// (slimm_amd/distributed.py is the same flow over torch.distributed) are collectives enqueued on the members' own HIP
// streams:
//     ncclAllGather  of the coverage summaries   (slimm_coverage_summary -> slimm_finish_coverage_merged)
//     ncclAllReduce  of the partial results       (slimm_filter_alignments_launch -> slimm_install_merged_partials)
// RCCL is loaded with dlopen when a group of distinct devices is created (the library itself does not link it: a
// process that also runs torch has torch's copy of RCCL, and a single-GPU user needs none).  When RCCL is not there --
// or the same device is named several times, which is how the tests run a group on one GPU -- the same two collectives
// are device-to-device copies ordered by events plus a summing kernel.  The reference has no counterpart (one process,
// one thread, no GPU).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/slimm_hip.h"
#include "force.h"

namespace {

constexpr uint64_t kKeyMask = (1ull << 62) - 1;  // include/slimm_hip.h: the significant bits of a read key

// ---- the five RCCL entry points this file uses, by their documented C signatures (rccl.h)
struct Rccl {
    typedef struct ncclComm* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;  // (optional: the all-to-all of the sliced form)
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    static constexpr int kInt32 = 2, kSum = 0;  // ncclInt32, ncclSum
    void* handle = nullptr;
    bool load() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(handle, "ncclAllGather"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(handle, "ncclAllReduce"));
        Send = reinterpret_cast<decltype(Send)>(dlsym(handle, "ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(dlsym(handle, "ncclRecv"));
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllGather && AllReduce;
    }
};

// ncclGroupStart ... ncclGroupEnd with the end on every way out (an error between the two must not leave the thread's
// RCCL group open: the ncclCommDestroy of slimm_group_destroy would run inside it)
struct RcclGroup {
    Rccl& r;
    bool open;
    explicit RcclGroup(Rccl& rccl) : r(rccl), open(rccl.GroupStart() == 0) {}
    int end() {
        if (!open) return -1;
        open = false;
        return r.GroupEnd();
    }
    ~RcclGroup() {
        if (open) (void)r.GroupEnd();
    }
};

// out[w] = sum over the n stretches of `parts` (each `words` long): the all-reduce of the copy form
__global__ __launch_bounds__(256) void k_group_sum(const uint32_t* __restrict__ parts, uint32_t n, uint32_t words,
                                                   uint32_t* __restrict__ out) {
    for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < words; w += gridDim.x * 256) {
        uint32_t s = 0;
        for (uint32_t j = 0; j < n; ++j) s += parts[static_cast<size_t>(j) * words + w];
        out[w] = s;
    }
}

// cuts[i] = the first qName-run start at or behind record i * n / G of run-marked records (n when there is none): the
// members' stretches of a file that member 0 decoded (partition.py: contiguous_cuts, on the device)
__global__ __launch_bounds__(64) void k_group_cuts(const uint32_t* __restrict__ word, uint64_t n, uint32_t G, uint64_t* __restrict__ cuts) {
    for (uint32_t i = threadIdx.x; i <= G; i += 64u) {
        uint64_t c = i == G ? n : (static_cast<uint64_t>(i) * n) / G;
        while (c != 0 && c < n && (word[c] >> 31) == 0u) ++c;
        cuts[i] = c;
    }
}

}  // namespace

struct slimm_group {
    std::vector<slimm_ctx*> ctx;
    std::vector<int> device;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> ready, copied;   // per member: "my buffer is written" / "I have read everybody's buffer"
    std::vector<uint32_t*> dealt_word;       // per member: its stretch of a file member 0 decoded (deal_from_member0)
    std::vector<int32_t*> dealt_pos;
    std::vector<size_t> dealt_cap;
    std::vector<uint32_t*> scratch;          // per member: n x words receive buffer
    std::vector<size_t> scratch_words;
    std::vector<Rccl::comm_t> comm;
    Rccl rccl;
    bool use_rccl = false;
    int order = SLIMM_ORDER_GROUPED;
    // dealing records by read
    uint32_t cur = 0;                        // member that takes the next stretch of a grouped stream
    std::vector<uint64_t> carry_key;         // the last qName run of the batch before: it may go on in the next batch
    std::vector<int32_t> carry_ref, carry_pos;
    std::vector<uint16_t> carry_flag;
    std::vector<uint32_t> carry_check;       // (streams pushed with check words)
    std::vector<uint32_t> carry_word;        // (run-marked records: the words; their positions in carry_pos)
    int checked = -1;                        // -1: nothing pushed yet, 0 / 1: the file's pushes carry no / carry check words
    int exchange = SLIMM_EXCHANGE_AUTO;
    uint32_t n_refs = 0;
    bool have_last = false;
    uint64_t last_key = 0;
    std::string err;
};

namespace {

int gfail(slimm_group* g, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    return code;
}
std::string g_group_create_error;

int member_failed(slimm_group* g, uint32_t i, int rc, const char* what) {
    return gfail(g, rc, "member %u (device %d): %s: %s", i, g->device[i], what, slimm_last_error(g->ctx[i]));
}

#define GTRY(g, i, call)                                       \
    do {                                                       \
        int rc_ = (call);                                      \
        if (rc_ < 0) return member_failed(g, i, rc_, #call);   \
    } while (0)
#define GHIP(g, expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return gfail(g, SLIMM_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

uint32_t ref_count(slimm_group* g) { return g->n_refs; }

int ensure_scratch(slimm_group* g, uint32_t i, size_t words) {
    if (g->scratch_words[i] >= words) return SLIMM_OK;
    GHIP(g, hipSetDevice(g->device[i]));
    if (g->scratch[i]) (void)hipFree(g->scratch[i]);
    g->scratch[i] = nullptr;
    g->scratch_words[i] = 0;
    GHIP(g, hipMalloc(reinterpret_cast<void**>(&g->scratch[i]), words * 4));
    g->scratch_words[i] = words;
    return SLIMM_OK;
}

// nobody goes on (and overwrites a buffer its peers were given to read) before every member's copies are done: each
// member records "I have read everybody's buffer", each member's stream waits for all of them
int read_fence(slimm_group* g) {
    const uint32_t n = static_cast<uint32_t>(g->ctx.size());
    for (uint32_t i = 0; i < n; ++i) {
        GHIP(g, hipSetDevice(g->device[i]));
        GHIP(g, hipEventRecord(g->copied[i], g->stream[i]));
    }
    for (uint32_t i = 0; i < n; ++i) {
        GHIP(g, hipSetDevice(g->device[i]));
        for (uint32_t j = 0; j < n; ++j)
            if (j != i) GHIP(g, hipStreamWaitEvent(g->stream[i], g->copied[j], 0));
    }
    return SLIMM_OK;
}

// recv[i] (on member i's device, n x words) = send[0] | send[1] | ... | send[n-1], enqueued on the members' streams
int all_gather(slimm_group* g, const std::vector<uint32_t*>& send, size_t words, std::vector<uint32_t*>& recv) {
    const uint32_t n = static_cast<uint32_t>(g->ctx.size());
    recv.assign(n, nullptr);
    for (uint32_t i = 0; i < n; ++i) {
        int rc = ensure_scratch(g, i, n * words);
        if (rc != SLIMM_OK) return rc;
        recv[i] = g->scratch[i];
    }
    if (g->use_rccl) {
        RcclGroup grp(g->rccl);
        if (!grp.open) return gfail(g, SLIMM_E_HIP, "ncclGroupStart failed");
        for (uint32_t i = 0; i < n; ++i) {
            GHIP(g, hipSetDevice(g->device[i]));
            if (g->rccl.AllGather(send[i], recv[i], words, Rccl::kInt32, g->comm[i], g->stream[i]) != 0)
                return gfail(g, SLIMM_E_HIP, "ncclAllGather failed on member %u", i);
        }
        if (grp.end() != 0) return gfail(g, SLIMM_E_HIP, "ncclGroupEnd failed");
        return SLIMM_OK;
    }
    for (uint32_t j = 0; j < n; ++j) {
        GHIP(g, hipSetDevice(g->device[j]));
        GHIP(g, hipEventRecord(g->ready[j], g->stream[j]));
    }
    for (uint32_t i = 0; i < n; ++i) {
        GHIP(g, hipSetDevice(g->device[i]));
        for (uint32_t j = 0; j < n; ++j) {
            if (j != i) GHIP(g, hipStreamWaitEvent(g->stream[i], g->ready[j], 0));
            GHIP(g, hipMemcpyAsync(recv[i] + static_cast<size_t>(j) * words, send[j], words * 4, hipMemcpyDeviceToDevice,
                                   g->stream[i]));
        }
    }
    return read_fence(g);
}

// recv[i] = chunk i of send[0] | chunk i of send[1] | ... (chunks of `words` each): every member gets ITS chunk of every
// member's buffer -- ncclSend / ncclRecv pairs inside one RCCL group (what ncclAllToAll is made of), or copies
int all_to_all(slimm_group* g, const std::vector<uint32_t*>& send, size_t words, std::vector<uint32_t*>& recv) {
    const uint32_t n = static_cast<uint32_t>(g->ctx.size());
    recv.assign(n, nullptr);
    for (uint32_t i = 0; i < n; ++i) {
        int rc = ensure_scratch(g, i, n * words);
        if (rc != SLIMM_OK) return rc;
        recv[i] = g->scratch[i];
    }
    if (g->use_rccl) {
        RcclGroup grp(g->rccl);
        if (!grp.open) return gfail(g, SLIMM_E_HIP, "ncclGroupStart failed");
        for (uint32_t i = 0; i < n; ++i) {
            GHIP(g, hipSetDevice(g->device[i]));
            for (uint32_t j = 0; j < n; ++j) {
                if (g->rccl.Send(send[i] + static_cast<size_t>(j) * words, words, Rccl::kInt32, static_cast<int>(j), g->comm[i],
                                 g->stream[i]) != 0 ||
                    g->rccl.Recv(recv[i] + static_cast<size_t>(j) * words, words, Rccl::kInt32, static_cast<int>(j), g->comm[i],
                                 g->stream[i]) != 0)
                    return gfail(g, SLIMM_E_HIP, "ncclSend / ncclRecv failed on member %u", i);
            }
        }
        if (grp.end() != 0) return gfail(g, SLIMM_E_HIP, "ncclGroupEnd failed");
        return SLIMM_OK;
    }
    for (uint32_t j = 0; j < n; ++j) {
        GHIP(g, hipSetDevice(g->device[j]));
        GHIP(g, hipEventRecord(g->ready[j], g->stream[j]));
    }
    for (uint32_t i = 0; i < n; ++i) {
        GHIP(g, hipSetDevice(g->device[i]));
        for (uint32_t j = 0; j < n; ++j) {
            if (j != i) GHIP(g, hipStreamWaitEvent(g->stream[i], g->ready[j], 0));
            GHIP(g, hipMemcpyAsync(recv[i] + static_cast<size_t>(j) * words, send[j] + static_cast<size_t>(i) * words, words * 4,
                                   hipMemcpyDeviceToDevice, g->stream[i]));
        }
    }
    return read_fence(g);
}

// buf[i][w] = sum over the members of buf[j][w], in place, enqueued on the members' streams
int all_reduce_sum(slimm_group* g, const std::vector<uint32_t*>& buf, size_t words) {
    const uint32_t n = static_cast<uint32_t>(g->ctx.size());
    if (g->use_rccl) {
        RcclGroup grp(g->rccl);
        if (!grp.open) return gfail(g, SLIMM_E_HIP, "ncclGroupStart failed");
        for (uint32_t i = 0; i < n; ++i) {
            GHIP(g, hipSetDevice(g->device[i]));
            if (g->rccl.AllReduce(buf[i], buf[i], words, Rccl::kInt32, Rccl::kSum, g->comm[i], g->stream[i]) != 0)
                return gfail(g, SLIMM_E_HIP, "ncclAllReduce failed on member %u", i);
        }
        if (grp.end() != 0) return gfail(g, SLIMM_E_HIP, "ncclGroupEnd failed");
        return SLIMM_OK;
    }
    // the copy form goes in pieces, so that the receive buffers stay small whatever is reduced (the bins form sums
    // 2 Bp + 16 words: 160 MB at config 2); all_gather ends with the fence that lets the sums overwrite the buffers
    const size_t piece = 16u << 20;  // words
    for (size_t lo = 0; lo < words; lo += piece) {
        const size_t w = std::min(piece, words - lo);
        std::vector<uint32_t*> part(n), parts;
        for (uint32_t i = 0; i < n; ++i) part[i] = buf[i] + lo;
        int rc = all_gather(g, part, w, parts);  // parts[i] = everybody's piece, on member i's device
        if (rc != SLIMM_OK) return rc;
        for (uint32_t i = 0; i < n; ++i) {
            GHIP(g, hipSetDevice(g->device[i]));
            const uint32_t blocks = static_cast<uint32_t>(std::min<size_t>(1024, (w + 255) / 256));
            hipLaunchKernelGGL(k_group_sum, dim3(std::max(1u, blocks)), dim3(256), 0, g->stream[i], parts[i], n,
                               static_cast<uint32_t>(w), part[i]);
        }
        if (lo + piece < words) {  // (the next piece reuses the receive buffers)
            int rf = read_fence(g);
            if (rf != SLIMM_OK) return rf;
        }
    }
    return SLIMM_OK;
}

int push_to(slimm_group* g, uint32_t i, const uint64_t* key, const int32_t* ref, const int32_t* pos, const uint16_t* flag,
            const uint32_t* check, uint64_t n) {
    if (n == 0) return SLIMM_OK;
    if (!flag)  // packed records: the flag bits ride in the key (slimm_pack_key)
        GTRY(g, i, slimm_push_records_packed(g->ctx[i], key, ref, pos, n));
    else if (check)
        GTRY(g, i, slimm_push_records_checked(g->ctx[i], key, ref, pos, flag, check, n));
    else
        GTRY(g, i, slimm_push_records(g->ctx[i], key, ref, pos, flag, n));
    return SLIMM_OK;
}

int flush_carry(slimm_group* g, uint32_t to) {
    if (!g->carry_word.empty()) {
        GTRY(g, to, slimm_push_records_marked(g->ctx[to], g->carry_word.data(), g->carry_pos.data(), g->carry_word.size()));
        g->carry_word.clear();
        g->carry_pos.clear();
        return SLIMM_OK;
    }
    if (g->carry_key.empty()) return SLIMM_OK;
    int rc = push_to(g, to, g->carry_key.data(), g->carry_ref.data(), g->carry_pos.data(),
                     g->carry_flag.empty() ? nullptr : g->carry_flag.data(),
                     g->carry_check.empty() ? nullptr : g->carry_check.data(), g->carry_key.size());
    g->carry_key.clear();
    g->carry_ref.clear();
    g->carry_pos.clear();
    g->carry_flag.clear();
    g->carry_check.clear();
    return rc;
}

// Deals a batch to the members by read.  Name-grouped streams: the batch up to its last qName-run start goes to the
// current member (together with the run the batch before ended in), the last run waits for the next batch -- it may go
// on there -- and the next member is up: contiguous stretches of the file, cut at run boundaries, in turn.  Any other
// order: member = key mod n (two names that collide in the key land on the same member, so check words do their work).
int deal(slimm_group* g, const uint64_t* key, const int32_t* ref, const int32_t* pos, const uint16_t* flag, const uint32_t* check,
         uint64_t n) {
    if (!g) return SLIMM_E_INVALID;
    if (n == 0) return SLIMM_OK;
    if (!key || !ref || !pos) return gfail(g, SLIMM_E_INVALID, "null record array");
    const int form = !flag ? 2 : (check ? 1 : 0);  // packed (no flag array: the bits ride in the key) / checked / plain
    if (g->checked >= 0 && g->checked != form)
        return gfail(g, SLIMM_E_INVALID, "packed, checked and plain pushes do not mix within a file");
    g->checked = form;
    const uint64_t kMask = flag ? kKeyMask : (1ull << 61) - 1ull;  // the identity bits of a key
    const uint32_t m = static_cast<uint32_t>(g->ctx.size());
    if (m == 1) return push_to(g, 0, key, ref, pos, flag, check, n);
    if (g->order != SLIMM_ORDER_GROUPED) {
        std::vector<std::vector<uint64_t>> k(m);
        std::vector<std::vector<int32_t>> r(m), p(m);
        std::vector<std::vector<uint16_t>> f(m);
        std::vector<std::vector<uint32_t>> c(m);
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t o = static_cast<uint32_t>((key[i] & kMask) % m);
            k[o].push_back(key[i]);
            r[o].push_back(ref[i]);
            p[o].push_back(pos[i]);
            if (flag) f[o].push_back(flag[i]);
            if (check) c[o].push_back(check[i]);
        }
        for (uint32_t o = 0; o < m; ++o) {
            int rc = push_to(g, o, k[o].data(), r[o].data(), p[o].data(), flag ? f[o].data() : nullptr,
                             check ? c[o].data() : nullptr, k[o].size());
            if (rc != SLIMM_OK) return rc;
        }
        return SLIMM_OK;
    }
    // the run the batch before ended in goes where this batch's head goes
    int rc = flush_carry(g, g->cur);
    if (rc != SLIMM_OK) return rc;
    uint64_t last_start = n;  // index of the batch's last run start, n = none found
    for (uint64_t i = n; i-- > 1;)
        if ((key[i] ^ key[i - 1]) & kMask) {
            last_start = i;
            break;
        }
    if (last_start == n && !(g->have_last && ((key[0] ^ g->last_key) & kMask) == 0)) last_start = 0;  // one run, a new one
    g->have_last = true;
    g->last_key = key[n - 1];
    if (last_start == n || last_start == 0) {
        // no boundary inside the batch: it stays with the current member, and so does whatever continues it
        return push_to(g, g->cur, key, ref, pos, flag, check, n);
    }
    rc = push_to(g, g->cur, key, ref, pos, flag, check, last_start);
    if (rc != SLIMM_OK) return rc;
    g->carry_key.assign(key + last_start, key + n);
    g->carry_ref.assign(ref + last_start, ref + n);
    g->carry_pos.assign(pos + last_start, pos + n);
    if (flag) g->carry_flag.assign(flag + last_start, flag + n);
    if (check) g->carry_check.assign(check + last_start, check + n);
    g->cur = (g->cur + 1) % m;
    return SLIMM_OK;
}

// Run-marked records: the same contiguous stretches, cut where a word says a run starts.
int deal_marked(slimm_group* g, const uint32_t* word, const int32_t* pos, uint64_t n) {
    if (!g) return SLIMM_E_INVALID;
    if (n == 0) return SLIMM_OK;
    if (!word || !pos) return gfail(g, SLIMM_E_INVALID, "null record array");
    if (g->order != SLIMM_ORDER_GROUPED)
        return gfail(g, SLIMM_E_INVALID, "run-marked records carry no read identity: the group must be created for input grouped by name");
    if (g->checked >= 0 && g->checked != 3)
        return gfail(g, SLIMM_E_INVALID, "run-marked and other pushes do not mix within a file");
    g->checked = 3;
    const uint32_t m = static_cast<uint32_t>(g->ctx.size());
    if (m == 1) {
        GTRY(g, 0, slimm_push_records_marked(g->ctx[0], word, pos, n));
        return SLIMM_OK;
    }
    int rc = flush_carry(g, g->cur);  // the run the batch before ended in goes where this batch's head goes
    if (rc != SLIMM_OK) return rc;
    uint64_t last_start = 0;  // index of the batch's last run start behind its first record, 0 = none
    for (uint64_t i = n; i-- > 1;)
        if (word[i] >> 31) {
            last_start = i;
            break;
        }
    if (last_start == 0) {  // no boundary inside the batch: it stays with the current member, and so does what continues it
        GTRY(g, g->cur, slimm_push_records_marked(g->ctx[g->cur], word, pos, n));
        return SLIMM_OK;
    }
    GTRY(g, g->cur, slimm_push_records_marked(g->ctx[g->cur], word, pos, last_start));
    g->carry_word.assign(word + last_start, word + n);
    g->carry_pos.assign(pos + last_start, pos + n);
    g->cur = (g->cur + 1) % m;
    return SLIMM_OK;
}

}  // namespace

extern "C" {

int slimm_group_create(const slimm_config* cfg, const int* devices, uint32_t n_devices, slimm_group** out) {
    if (!cfg || !devices || !out || n_devices == 0 || n_devices > 255) {
        g_group_create_error = "slimm_group_create: 1 .. 255 devices";
        return SLIMM_E_INVALID;
    }
    slimm_group* g = new slimm_group();
    g->order = cfg->record_order;
    g->n_refs = cfg->n_refs;
    bool distinct = true;
    for (uint32_t i = 0; i < n_devices; ++i)
        for (uint32_t j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
    // the members' contexts side by side, a thread each: a device's part of the HIP runtime starts with its first context,
    // and the tables of eight members one after the other are eight times one member's
    std::vector<slimm_ctx*> made(n_devices, nullptr);
    std::vector<int> made_rc(n_devices, SLIMM_OK);
    std::vector<std::string> made_err(n_devices);
    {
        std::vector<std::thread> th;
        for (uint32_t i = 0; i < n_devices; ++i)
            th.emplace_back([&, i] {
                slimm_config c = *cfg;
                c.device = devices[i];
                made_rc[i] = slimm_create(&c, &made[i]);
                if (made_rc[i] != SLIMM_OK) made_err[i] = slimm_last_error(nullptr);
            });
        for (auto& t : th) t.join();
    }
    for (uint32_t i = 0; i < n_devices; ++i)
        if (made_rc[i] != SLIMM_OK) {
            g_group_create_error = std::string("member ") + std::to_string(i) + ": " + made_err[i];
            for (slimm_ctx* c : made)
                if (c) slimm_destroy(c);
            slimm_group_destroy(g);
            return made_rc[i];
        }
    for (uint32_t i = 0; i < n_devices; ++i) {
        slimm_ctx* ctx = made[i];
        g->ctx.push_back(ctx);
        g->device.push_back(devices[i]);
        void* s = nullptr;
        (void)slimm_get_stream(ctx, &s);
        g->stream.push_back(static_cast<hipStream_t>(s));
        g->scratch.push_back(nullptr);
        g->scratch_words.push_back(0);
        g->dealt_word.push_back(nullptr);
        g->dealt_pos.push_back(nullptr);
        g->dealt_cap.push_back(0);
        hipEvent_t a = nullptr, b = nullptr;
        (void)hipSetDevice(devices[i]);
        if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
            g_group_create_error = "hipEventCreate failed";
            for (uint32_t j = i + 1; j < n_devices; ++j) slimm_destroy(made[j]);  // (the group does not hold them yet)
            slimm_group_destroy(g);
            return SLIMM_E_HIP;
        }
        g->ready.push_back(a);
        g->copied.push_back(b);
    }
    // collectives: RCCL over xGMI for distinct devices (SLIMM_FORCE="group_collectives=copy|rccl" overrides the choice: force.h)
    const char* how = nullptr;
    (void)slimm::forced_text("group_collectives", &how);
    const bool how_rccl = how && strncmp(how, "rccl", 4) == 0;
    const bool want_rccl = how ? how_rccl : (distinct && n_devices > 1);
    if (want_rccl && (distinct || n_devices == 1) && g->rccl.load()) {
        g->comm.assign(n_devices, nullptr);
        if (g->rccl.CommInitAll(g->comm.data(), static_cast<int>(n_devices), g->device.data()) == 0) {
            g->use_rccl = true;
        } else {
            g->comm.clear();
        }
    }
    if (how_rccl && !g->use_rccl) {
        g_group_create_error = "SLIMM_FORCE group_collectives=rccl, but librccl could not be loaded or initialised for these devices";
        slimm_group_destroy(g);
        return SLIMM_E_HIP;
    }
    if (n_devices > 1 || g->use_rccl)  // the collectives run on the members' streams: no host fences around them
        for (slimm_ctx* c : g->ctx) (void)slimm_set_stream_ordered(c, 1);
    *out = g;
    return SLIMM_OK;
}

void slimm_group_destroy(slimm_group* g) {
    if (!g) return;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->device[i]);
        if (g->stream[i]) (void)hipStreamSynchronize(g->stream[i]);
    }
    if (g->use_rccl)
        for (Rccl::comm_t c : g->comm)
            if (c) (void)g->rccl.CommDestroy(c);
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        (void)hipSetDevice(g->device[i]);
        if (i < g->scratch.size() && g->scratch[i]) (void)hipFree(g->scratch[i]);
        if (i < g->dealt_word.size() && g->dealt_word[i]) (void)hipFree(g->dealt_word[i]);
        if (i < g->dealt_pos.size() && g->dealt_pos[i]) (void)hipFree(g->dealt_pos[i]);
        if (i < g->ready.size() && g->ready[i]) (void)hipEventDestroy(g->ready[i]);
        if (i < g->copied.size() && g->copied[i]) (void)hipEventDestroy(g->copied[i]);
        slimm_destroy(g->ctx[i]);
    }
    delete g;
}

const char* slimm_group_last_error(const slimm_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }
uint32_t slimm_group_size(const slimm_group* g) { return g ? static_cast<uint32_t>(g->ctx.size()) : 0u; }
slimm_ctx* slimm_group_context(slimm_group* g, uint32_t i) { return (g && i < g->ctx.size()) ? g->ctx[i] : nullptr; }
int slimm_group_uses_rccl(const slimm_group* g) { return (g && g->use_rccl) ? 1 : 0; }

int slimm_group_reset(slimm_group* g) {
    if (!g) return SLIMM_E_INVALID;
    for (uint32_t i = 0; i < g->ctx.size(); ++i) GTRY(g, i, slimm_reset(g->ctx[i]));
    g->cur = 0;
    g->carry_key.clear();
    g->carry_ref.clear();
    g->carry_pos.clear();
    g->carry_flag.clear();
    g->carry_check.clear();
    g->carry_word.clear();
    g->checked = -1;
    g->have_last = false;
    return SLIMM_OK;
}

int slimm_group_push_records(slimm_group* g, const uint64_t* key, const int32_t* ref, const int32_t* pos, const uint16_t* flag,
                             uint64_t n) {
    if (g && n && !flag) return gfail(g, SLIMM_E_INVALID, "null record array");
    return deal(g, key, ref, pos, flag, nullptr, n);
}
int slimm_group_push_records_checked(slimm_group* g, const uint64_t* key, const int32_t* ref, const int32_t* pos,
                                     const uint16_t* flag, const uint32_t* check, uint64_t n) {
    if (g && n && (!check || !flag)) return gfail(g, SLIMM_E_INVALID, "null record array");
    return deal(g, key, ref, pos, flag, check, n);
}
int slimm_group_push_records_packed(slimm_group* g, const uint64_t* packed_key, const int32_t* ref, const int32_t* pos, uint64_t n) {
    return deal(g, packed_key, ref, pos, nullptr, nullptr, n);
}
int slimm_group_push_records_marked(slimm_group* g, const uint32_t* word, const int32_t* pos, uint64_t n) {
    return deal_marked(g, word, pos, n);
}
int slimm_group_set_exchange(slimm_group* g, int mode) {
    if (!g || mode < SLIMM_EXCHANGE_AUTO || mode > SLIMM_EXCHANGE_BINS) return SLIMM_E_INVALID;
    g->exchange = mode;
    return SLIMM_OK;
}
int slimm_group_exchange(const slimm_group* g) {  // what AUTO means for this group
    if (!g) return SLIMM_E_INVALID;
    if (g->exchange != SLIMM_EXCHANGE_AUTO) return g->exchange;
    const bool can_slice = !g->use_rccl || (g->rccl.Send && g->rccl.Recv);
    return (g->ctx.size() > 2 && can_slice) ? SLIMM_EXCHANGE_SLICED : SLIMM_EXCHANGE_SUMMARY;
}

// Member 0 holds a GROUPED file it decoded itself (its windows were pushed to slimm_group_context(g, 0)) and nothing was
// dealt through slimm_group_push_records*: the members' stretches are cut at qName-run starts on the device and copied device to
// device -- 8 bytes per record; member 0 keeps the first one where it lies.
static int deal_from_member0(slimm_group* g) {
    const uint32_t G = static_cast<uint32_t>(g->ctx.size());
    const uint64_t* key = nullptr;
    const int32_t *ref = nullptr, *pos = nullptr;
    const uint16_t* flag = nullptr;
    uint64_t n = 0;
    int form = 0;
    GTRY(g, 0, slimm_records_device(g->ctx[0], &key, &ref, &pos, &flag, &n, &form));
    if (n == 0 || form != 2) return SLIMM_OK;   // (nothing there, or not run-marked records: as pushed)
    for (uint32_t i = 1; i < G; ++i) {
        uint64_t ni = 0;
        int fi = 0;
        const uint64_t* k2;
        const int32_t *r2, *p2;
        const uint16_t* f2;
        GTRY(g, i, slimm_records_device(g->ctx[i], &k2, &r2, &p2, &f2, &ni, &fi));
        if (ni) return SLIMM_OK;                // (the caller dealt records itself)
    }
    GHIP(g, hipSetDevice(g->device[0]));
    uint64_t* d_cuts = nullptr;
    GHIP(g, hipMalloc(reinterpret_cast<void**>(&d_cuts), (G + 1) * sizeof(uint64_t)));
    hipLaunchKernelGGL(k_group_cuts, dim3(1), dim3(64), 0, g->stream[0], reinterpret_cast<const uint32_t*>(ref), n, G, d_cuts);
    std::vector<uint64_t> cuts(G + 1);
    hipError_t e = hipMemcpyAsync(cuts.data(), d_cuts, (G + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(g->stream[0]);
    (void)hipFree(d_cuts);
    if (e != hipSuccess) return gfail(g, SLIMM_E_HIP, "deal_from_member0: %s", hipGetErrorString(e));
    for (uint32_t i = 1; i <= G; ++i) cuts[i] = std::max(cuts[i], cuts[i - 1]);
    for (uint32_t i = 1; i < G; ++i) {
        const uint64_t lo = cuts[i], ni = cuts[i + 1] - cuts[i];
        GHIP(g, hipSetDevice(g->device[i]));
        if (g->dealt_cap[i] < ni) {
            if (g->dealt_word[i]) (void)hipFree(g->dealt_word[i]);
            if (g->dealt_pos[i]) (void)hipFree(g->dealt_pos[i]);
            g->dealt_word[i] = nullptr;
            g->dealt_pos[i] = nullptr;
            g->dealt_cap[i] = 0;
            GHIP(g, hipMalloc(reinterpret_cast<void**>(&g->dealt_word[i]), (ni + 16) * 4));
            GHIP(g, hipMalloc(reinterpret_cast<void**>(&g->dealt_pos[i]), (ni + 16) * 4));
            g->dealt_cap[i] = ni;
        }
        if (ni) {
            GHIP(g, hipMemcpyPeerAsync(g->dealt_word[i], g->device[i], ref + lo, g->device[0], ni * 4, g->stream[i]));
            GHIP(g, hipMemcpyPeerAsync(g->dealt_pos[i], g->device[i], pos + lo, g->device[0], ni * 4, g->stream[i]));
        }
        GTRY(g, i, slimm_set_records_device_marked(g->ctx[i], g->dealt_word[i], g->dealt_pos[i], ni));
    }
    // (member 0 goes on only when its peers have read their stretches: its arrays are its own again after the file)
    for (uint32_t i = 1; i < G; ++i) {
        GHIP(g, hipSetDevice(g->device[i]));
        GHIP(g, hipEventRecord(g->copied[i], g->stream[i]));
    }
    GHIP(g, hipSetDevice(g->device[0]));
    for (uint32_t i = 1; i < G; ++i) GHIP(g, hipStreamWaitEvent(g->stream[0], g->copied[i], 0));
    GTRY(g, 0, slimm_set_records_device_marked(g->ctx[0], reinterpret_cast<const uint32_t*>(ref), pos, cuts[1]));
    return SLIMM_OK;
}

// slimm::get_profiles() (src/slimm.hpp:447-489) over the members' reads: phases A, B, C(1) on every member with the two
// exchanges in between, propagation and the profile on member 0 (every member holds the same merged results).
int slimm_group_get_profiles(slimm_group* g, const char* path) {
    if (!g) return SLIMM_E_INVALID;
    const uint32_t n = static_cast<uint32_t>(g->ctx.size());
    int rc = flush_carry(g, g->cur);
    if (rc != SLIMM_OK) return rc;
    if (n > 1 && g->order == SLIMM_ORDER_GROUPED) {
        rc = deal_from_member0(g);
        if (rc != SLIMM_OK) return rc;
    }
    if (n == 1 && !g->use_rccl) {  // (a group of one with RCCL forced goes the long way: the test of the RCCL calls)
        rc = slimm_get_profiles(g->ctx[0], path);
        if (rc < 0) return member_failed(g, 0, rc, "slimm_get_profiles");
        return rc;
    }
    // ---- phase A on every member, then exchange 1 in one of three forms (slimm_amd/distributed.py has the same three):
    //   SUMMARY  ncclAllGather of [per-reference sums | scalars | one bit per bin]
    //   SLICED   all-to-all of the bitmaps cut into one slice per member (ncclSend / ncclRecv in one group), every member
    //            merges ITS slice of everybody's bitmaps, then a small ncclAllReduce of [sums, partial counts | scalars]
    //   BINS     the literal ncclAllReduce(ncclSum) over [cov | uniq_cov | scalars]: every member then holds the global
    //            coverage arrays (what the reference's -ro / -co outputs read, src/slimm.hpp:846-943)
    const int how = slimm_group_exchange(g);
    for (uint32_t i = 0; i < n; ++i)
        GTRY(g, i, slimm_prepare_summary(g->ctx[i], how == SLIMM_EXCHANGE_SLICED ? n : (how == SLIMM_EXCHANGE_SUMMARY ? 1u : 0u)));
    for (uint32_t i = 0; i < n; ++i) GTRY(g, i, slimm_analyze_alignments(g->ctx[i]));
    bool hits = false;
    if (how == SLIMM_EXCHANGE_BINS) {
        std::vector<uint32_t*> bins(n);
        uint64_t words = 0;
        for (uint32_t i = 0; i < n; ++i) {
            void* p = nullptr;
            uint64_t w = 0;
            GTRY(g, i, slimm_coverage_buffer(g->ctx[i], &p, &w));
            bins[i] = static_cast<uint32_t*>(p);
            if (i && w != words) return gfail(g, SLIMM_E_INVALID, "members disagree about the coverage buffer size");
            words = w;
        }
        rc = all_reduce_sum(g, bins, words);
        if (rc != SLIMM_OK) return rc;
        for (uint32_t i = 0; i < n; ++i) {
            rc = slimm_finish_coverage(g->ctx[i]);
            if (rc < 0) return member_failed(g, i, rc, "slimm_finish_coverage");
            hits = hits || rc != SLIMM_E_NO_HITS;
        }
    } else {
        std::vector<uint32_t*> mine(n), gathered;
        uint64_t words = 0;
        for (uint32_t i = 0; i < n; ++i) {
            void* p = nullptr;
            uint64_t w = 0;
            GTRY(g, i, slimm_coverage_summary(g->ctx[i], &p, &w));
            mine[i] = static_cast<uint32_t*>(p);
            if (i && w != words) return gfail(g, SLIMM_E_INVALID, "members disagree about the summary size");
            words = w;
        }
        if (how == SLIMM_EXCHANGE_SLICED) {
            const uint64_t head = 4ull * ref_count(g) + 16;  // [4 R sums | 16 scalars] in front of the n chunks
            if (words < head || (words - head) % n) return gfail(g, SLIMM_E_INVALID, "unexpected sliced summary size");
            const uint64_t chunk = (words - head) / n;
            std::vector<uint32_t*> chunks(n), vec(n);
            for (uint32_t i = 0; i < n; ++i) chunks[i] = mine[i] + head;
            rc = all_to_all(g, chunks, chunk, gathered);
            if (rc != SLIMM_OK) return rc;
            uint64_t vw = 0;
            for (uint32_t i = 0; i < n; ++i) {
                void* v = nullptr;
                GTRY(g, i, slimm_merge_summary_slices(g->ctx[i], gathered[i], n, i, &v, &vw));
                vec[i] = static_cast<uint32_t*>(v);
            }
            rc = all_reduce_sum(g, vec, vw);
            if (rc != SLIMM_OK) return rc;
            for (uint32_t i = 0; i < n; ++i) {
                rc = slimm_finish_coverage_reduced(g->ctx[i]);
                if (rc < 0) return member_failed(g, i, rc, "slimm_finish_coverage_reduced");
                hits = hits || rc != SLIMM_E_NO_HITS;
            }
        } else {
            rc = all_gather(g, mine, words, gathered);
            if (rc != SLIMM_OK) return rc;
            for (uint32_t i = 0; i < n; ++i) {
                rc = slimm_finish_coverage_merged(g->ctx[i], gathered[i], n);
                if (rc < 0) return member_failed(g, i, rc, "slimm_finish_coverage_merged");
                hits = hits || rc != SLIMM_E_NO_HITS;
            }
        }
    }
    if (!hits) return SLIMM_E_NO_HITS;
    // ---- phase B / C(1) with exchange 2: the additive partial results, all-reduced in place
    uint32_t total_pairs = 0;
    for (int round = 0;; ++round) {
        if (round > 16) return gfail(g, SLIMM_E_INVALID, "pair set still overflowing after 16 rounds");
        for (uint32_t i = 0; i < n; ++i) GTRY(g, i, slimm_filter_alignments_launch(g->ctx[i]));
        std::vector<uint32_t*> part(n);
        uint64_t pw = 0;
        for (uint32_t i = 0; i < n; ++i) {
            void* p = nullptr;
            GTRY(g, i, slimm_partials_buffer(g->ctx[i], &p, &pw));
            part[i] = static_cast<uint32_t*>(p);
        }
        rc = all_reduce_sum(g, part, pw);
        if (rc != SLIMM_OK) return rc;
        bool again = false;
        for (uint32_t i = 0; i < n; ++i) {
            uint32_t tp = 0;
            rc = slimm_install_merged_partials(g->ctx[i], &tp);
            if (rc < 0) return member_failed(g, i, rc, "slimm_install_merged_partials");
            again = again || rc == SLIMM_E_RETRY;
            total_pairs = tp;
        }
        if (!again) break;
    }
    if (total_pairs) {  // rare (Q4): the union of the members' (taxon, reference) pairs, through the host
        std::vector<uint64_t> all;
        std::vector<slimm_partials> parts(n);
        for (uint32_t i = 0; i < n; ++i) {
            GTRY(g, i, slimm_get_partials(g->ctx[i], &parts[i]));
            all.insert(all.end(), parts[i].pairs, parts[i].pairs + parts[i].n_pairs);
        }
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        for (uint32_t i = 0; i < n; ++i) {
            slimm_partials in = parts[i];
            // (the arrays slimm_get_partials returned are the member's own: copy what slimm_set_partials will overwrite)
            std::vector<uint32_t> u2(in.uniq_reads_count2, in.uniq_reads_count2 + in.n_refs),
                lca(in.lca_count, in.lca_count + in.n_taxa_dense), mk(in.level_marks, in.level_marks + in.n_refs);
            in.uniq_reads_count2 = u2.data();
            in.lca_count = lca.data();
            in.level_marks = mk.data();
            in.pairs = all.data();
            in.n_pairs = static_cast<uint32_t>(all.size());
            GTRY(g, i, slimm_set_partials(g->ctx[i], &in));
        }
    }
    if (how == SLIMM_EXCHANGE_BINS) {  // the third coverage array of the -co output: summed over the members as well
        std::vector<uint32_t*> u2(n);
        uint64_t words = 0;
        for (uint32_t i = 0; i < n; ++i) {
            void* p = nullptr;
            GTRY(g, i, slimm_uniq_cov2_buffer(g->ctx[i], &p, &words));
            u2[i] = static_cast<uint32_t*>(p);
        }
        rc = all_reduce_sum(g, u2, words);
        if (rc != SLIMM_OK) return rc;
        for (uint32_t i = 0; i < n; ++i) {  // (the host reads the arrays next: slimm_get_bins copies on the same stream)
            GHIP(g, hipSetDevice(g->device[i]));
            GHIP(g, hipStreamSynchronize(g->stream[i]));
        }
    }
    GTRY(g, 0, slimm_get_reads_lca_count(g->ctx[0]));
    if (path) GTRY(g, 0, slimm_write_abundance_file(g->ctx[0], path));
    return SLIMM_OK;
}

}  // extern "C"
