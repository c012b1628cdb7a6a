// libhr_comm.so — native transport of the row-tiled frame (include/hr_comm.h).
//
// RCCL back end: grouped ncclSend / ncclRecv between direct neighbours (2 of the 7 xGMI links of a GPU, messages of 0.2-1 MB:
// latency-bound) on a dedicated communication stream, fenced against the caller's compute stream with events in both directions,
// so that the exchange overlaps whatever the caller enqueues next.  librccl is dlopen'ed (a process that has imported PyTorch
// already holds one; loading a second copy next to it must be avoided, and single-GPU hosts need none).
// Loopback back end: every rank of a group lives in this process on one device; the "wire" is hipMemcpyAsync.  The two ranks of a
// boundary rendezvous in a process-wide table: whichever arrives second moves the rows of BOTH directions on its own compute
// stream after waiting for the peer's ready-event, and leaves a done-event the peer's hr_comm_wait() waits for.
#include "../../include/hr_comm.h"
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace hr { void set_last_error(const std::string& s); }

namespace {

#define CK_HIP(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) { hr::set_last_error(std::string(#expr) + " failed: " + hipGetErrorString(e_)); return HR_ERR_HIP; } \
    } while (0)

// ---- RCCL, resolved at run time ------------------------------------------------------------------------------------------
struct NcclId { char internal[HR_COMM_ID_BYTES]; };
typedef void* NcclComm;
struct Rccl
{
    void* lib = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
constexpr int kNcclInt8 = 0;   // ncclInt8 / ncclChar

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
            if ((r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!r.lib) return;
        auto sym = [&](const char* n) { return dlsym(r.lib, n); };
        r.GetUniqueId    = (int (*)(NcclId*))sym("ncclGetUniqueId");
        r.CommInitRank   = (int (*)(NcclComm*, int, NcclId, int))sym("ncclCommInitRank");
        r.CommDestroy    = (int (*)(NcclComm))sym("ncclCommDestroy");
        r.GroupStart     = (int (*)())sym("ncclGroupStart");
        r.GroupEnd       = (int (*)())sym("ncclGroupEnd");
        r.Send           = (int (*)(const void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclSend");
        r.Recv           = (int (*)(void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclRecv");
        r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
    });
    return r;
}

#define CK_NCCL(expr)                                                                                                  \
    do {                                                                                                               \
        int e_ = (expr);                                                                                               \
        if (e_ != 0) { hr::set_last_error(std::string(#expr) + " failed: " + (rccl().GetErrorString ? rccl().GetErrorString(e_) : "nccl error")); return HR_ERR_COMM; } \
    } while (0)

// ---- loopback rendezvous ---------------------------------------------------------------------------------------------
// Ranks of a group are driven by ONE host thread in any interleaving (rank 0 may post the exchanges of three passes before rank 1
// posts its first): posts are queued per boundary side, the k-th post of the upper rank meets the k-th post of the lower rank.
// A collective a rank has posted.  wait_ticket() blocks (loopback only) until `open` reaches 0 — every side has met its partner — and
// then makes the caller's stream wait for the copies other ranks enqueued on its behalf.
struct TicketState
{
    int                     open = 0;   // boundary sides / gather rounds of this collective that have not met their partner yet
    std::vector<hipEvent_t> done;       // copies into / out of my images that sit on ANOTHER rank's stream
};
struct Post   // what one rank offers at one boundary for one exchange
{
    std::vector<hr_comm_image>   images;
    hipEvent_t                   ready = nullptr;   // recorded on the poster's compute stream: its rows are final
    int                          s0 = 0, s1 = 0;    // rows the poster sends (its own band rows next to the boundary)
    std::shared_ptr<TicketState> ticket;
};
struct Boundary { std::deque<Post> q[2]; };       // side 0 = upper rank
struct GatherPost { hr_comm_image image; hipEvent_t ready = nullptr; std::shared_ptr<TicketState> ticket; };
struct Group
{
    int world = 0, members = 0;
    std::vector<bool>                    joined;         // per rank
    std::vector<Boundary>                boundaries;     // world - 1
    std::vector<std::deque<GatherPost>>  gather;         // per rank
};
std::mutex                   g_mu;
std::condition_variable      g_cv;    // a pair / a gather round completed
std::map<std::string, Group> g_groups;

} // namespace

constexpr int kTicketRing = 64;

struct hr_comm
{
    hr_ctx*     ctx = nullptr;
    int         device = 0, world = 1, rank = 0;
    bool        loopback = false, joined = false;
    std::string name;
    NcclComm    nccl = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t  ev_compute = nullptr;                      // compute -> comm fence
    // tickets: every collective posted on this communicator gets the next number; wait_ticket(t) = everything up to t is complete
    int64_t     next_ticket = 1;
    hipEvent_t  ring[kTicketRing] = {};                    // RCCL: recorded on comm_stream after the collective of ticket t (slot t % ring)
    std::map<int64_t, std::shared_ptr<TicketState>> open;  // loopback: collectives not waited for yet
};

extern "C" hr_status hr_comm_destroy(hr_comm* c);
static hr_status comm_common(hr_ctx* ctx, int world, int rank, hr_comm** out, hr_comm*& c)
{
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world) { hr::set_last_error("hr_comm_create: invalid argument"); return HR_ERR_INVALID_ARG; }
    c = new (std::nothrow) hr_comm();
    if (!c) return HR_ERR_OUT_OF_MEMORY;
    c->ctx = ctx; c->device = hr_ctx_device(ctx); c->world = world; c->rank = rank;
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming);
    for (int i = 0; i < kTicketRing && e == hipSuccess; i++) e = hipEventCreateWithFlags(&c->ring[i], hipEventDisableTiming);
    if (e != hipSuccess) { hr::set_last_error(std::string("hr_comm_create: ") + hipGetErrorString(e)); hr_comm_destroy(c); c = nullptr; return HR_ERR_HIP; }
    return HR_OK;
}

extern "C" {

hr_status hr_comm_get_unique_id(uint8_t id[HR_COMM_ID_BYTES])
{
    if (!id) return HR_ERR_INVALID_ARG;
    if (!rccl().ok) { hr::set_last_error("librccl could not be loaded"); return HR_ERR_UNSUPPORTED; }
    NcclId nid;
    CK_NCCL(rccl().GetUniqueId(&nid));
    std::memcpy(id, nid.internal, HR_COMM_ID_BYTES);
    return HR_OK;
}

hr_status hr_comm_create_rccl(hr_ctx* ctx, int32_t world, int32_t rank, const uint8_t id[HR_COMM_ID_BYTES], hr_comm** out)
{
    if (!id) return HR_ERR_INVALID_ARG;
    if (!rccl().ok) { hr::set_last_error("librccl could not be loaded"); return HR_ERR_UNSUPPORTED; }
    hr_comm*  c;
    hr_status s = comm_common(ctx, world, rank, out, c);
    if (s != HR_OK) return s;
    NcclId nid;
    std::memcpy(nid.internal, id, HR_COMM_ID_BYTES);
    int e = rccl().CommInitRank(&c->nccl, world, nid, rank);
    if (e != 0)
    {
        hr::set_last_error(std::string("ncclCommInitRank failed: ") + (rccl().GetErrorString ? rccl().GetErrorString(e) : "nccl error"));
        hr_comm_destroy(c);
        return HR_ERR_COMM;
    }
    *out = c;
    return HR_OK;
}

hr_status hr_comm_create_loopback(hr_ctx* ctx, int32_t world, int32_t rank, const char* name, hr_comm** out)
{
    if (!name) return HR_ERR_INVALID_ARG;
    hr_comm*  c;
    hr_status s = comm_common(ctx, world, rank, out, c);
    if (s != HR_OK) return s;
    c->loopback = true; c->name = name;
    const char* why = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Group& g = g_groups[c->name];
        if (g.world == 0)
        {
            g.world = world;
            g.joined.assign(world, false);
            g.boundaries.resize(world > 1 ? world - 1 : 0);
            g.gather.resize(world);
        }
        if (g.world != world) why = "hr_comm_create_loopback: group exists with another world size";
        else if (g.joined[rank]) why = "hr_comm_create_loopback: this rank has already joined the group";
        else { g.joined[rank] = true; g.members++; c->joined = true; }
        if (why && g.members == 0) g_groups.erase(c->name);
    }
    if (why) { hr::set_last_error(why); hr_comm_destroy(c); return HR_ERR_INVALID_ARG; }   // one teardown path for every failure
    *out = c;
    return HR_OK;
}

hr_status hr_comm_destroy(hr_comm* c)
{
    if (!c) return HR_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->nccl && rccl().ok) (void)rccl().CommDestroy(c->nccl);
    if (c->loopback && c->joined)
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_groups.find(c->name);
        if (it != g_groups.end())
        {
            it->second.joined[c->rank] = false;
            if (--it->second.members <= 0)
            {
                for (Boundary& b : it->second.boundaries)
                    for (int k = 0; k < 2; k++)
                        for (Post& q : b.q[k]) if (q.ready) (void)hipEventDestroy(q.ready);
                for (auto& dq : it->second.gather)
                    for (GatherPost& q : dq) if (q.ready) (void)hipEventDestroy(q.ready);
                g_groups.erase(it);
            }
        }
    }
    for (auto& kv : c->open)
        for (hipEvent_t e : kv.second->done) (void)hipEventDestroy(e);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->ev_compute) (void)hipEventDestroy(c->ev_compute);
    for (hipEvent_t e : c->ring) if (e) (void)hipEventDestroy(e);
    delete c;
    return HR_OK;
}

int32_t hr_comm_rank(const hr_comm* c) { return c ? c->rank : -1; }
int32_t hr_comm_world(const hr_comm* c) { return c ? c->world : 0; }

// rows rank `r` sends towards boundary b (between rank b and b + 1), and where the received rows land
static void plan(const int32_t* bounds, int rows, int upper, int& up_s0, int& up_s1, int& lo_s0, int& lo_s1)
{
    // upper rank = `upper` (band bounds[upper] .. bounds[upper + 1]), lower rank = upper + 1; every band holds >= rows rows (checked)
    up_s0 = bounds[upper + 1] - rows; up_s1 = bounds[upper + 1];   // the upper rank's last rows
    lo_s0 = bounds[upper + 1]; lo_s1 = bounds[upper + 1] + rows;   // the lower rank's first rows
}

// loopback: retire the collectives that are kTicketRing tickets old and have met their partners (their done-events have long fired;
// a waiter that comes this late is ordered behind them by the newer entries of the same streams anyway, as with the RCCL ring)
static void prune(hr_comm* c)
{
    for (auto it = c->open.begin(); it != c->open.end() && it->first + kTicketRing < c->next_ticket;)
    {
        if (it->second->open > 0) { ++it; continue; }
        for (hipEvent_t e : it->second->done) (void)hipEventDestroy(e);
        it = c->open.erase(it);
    }
}

// RCCL: stamps the collective just enqueued on comm_stream with the next ticket
static hr_status stamp(hr_comm* c, hr_comm_ticket* ticket)
{
    const int64_t t = c->next_ticket++;
    CK_HIP(hipEventRecord(c->ring[t % kTicketRing], c->comm_stream));
    if (ticket) *ticket = t;
    return HR_OK;
}

hr_status hr_comm_exchange_rows(hr_comm* c, const hr_comm_image* images, int32_t n_images, const int32_t* bounds, int32_t rows, void* compute_stream_, hr_comm_ticket* ticket)
{
    if (ticket) *ticket = 0;
    if (!c || !images || n_images <= 0 || !bounds || rows <= 0) { hr::set_last_error("hr_comm_exchange_rows: invalid argument"); return HR_ERR_INVALID_ARG; }
    for (int r = 0; r < c->world; r++)
    {
        if (bounds[r + 1] <= bounds[r]) { hr::set_last_error("hr_comm_exchange_rows: band boundaries must ascend"); return HR_ERR_INVALID_ARG; }
        // the exchange talks to the two direct neighbours only: a band shorter than the apron would leave the apron rows owned by the
        // SECOND neighbour stale, and the bands would silently drift from the single-GPU image (tiling._TiledPass has the same guard)
        if (c->world > 1 && bounds[r + 1] - bounds[r] < rows)
        {
            hr::set_last_error("hr_comm_exchange_rows: band " + std::to_string(r) + " is shorter than the " + std::to_string(rows) + "-row apron; use fewer / taller bands");
            return HR_ERR_INVALID_ARG;
        }
    }
    if (c->world == 1) return HR_OK;
    hipStream_t cs = (hipStream_t)compute_stream_;
    CK_HIP(hipSetDevice(c->device));
    if (!c->loopback)
    {
        // compute -> comm fence, then one group with up to 2 x 2 x n_images point-to-point operations
        CK_HIP(hipEventRecord(c->ev_compute, cs));
        CK_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_compute, 0));
        CK_NCCL(rccl().GroupStart());
        int err = 0;   // inside the group every failure path must still close it (a dangling group hangs the next call of this thread)
        for (int side = 0; side < 2 && !err; side++)   // 0: boundary above me (peer rank - 1), 1: boundary below me (peer rank + 1)
        {
            const int upper = side == 0 ? c->rank - 1 : c->rank;
            if (upper < 0 || upper + 1 >= c->world) continue;
            int us0, us1, ls0, ls1;
            plan(bounds, rows, upper, us0, us1, ls0, ls1);
            const bool i_am_upper = side == 1;
            const int  peer = i_am_upper ? c->rank + 1 : c->rank - 1;
            const int  s0 = i_am_upper ? us0 : ls0, s1 = i_am_upper ? us1 : ls1, r0 = i_am_upper ? ls0 : us0, r1 = i_am_upper ? ls1 : us1;
            for (int i = 0; i < n_images && !err; i++)
            {
                char* base = (char*)images[i].data;
                const int64_t pitch = images[i].row_pitch_bytes;
                err = rccl().Send(base + (int64_t)s0 * pitch, (size_t)((int64_t)(s1 - s0) * pitch), kNcclInt8, peer, c->nccl, c->comm_stream);
                if (!err) err = rccl().Recv(base + (int64_t)r0 * pitch, (size_t)((int64_t)(r1 - r0) * pitch), kNcclInt8, peer, c->nccl, c->comm_stream);
            }
        }
        const int end = rccl().GroupEnd();
        if (err || end) { hr::set_last_error(std::string("hr_comm_exchange_rows: ") + (rccl().GetErrorString ? rccl().GetErrorString(err ? err : end) : "nccl error")); return HR_ERR_COMM; }
        return stamp(c, ticket);
    }
    // ---- loopback
    std::lock_guard<std::mutex> lk(g_mu);
    Group& g = g_groups[c->name];
    prune(c);   // a host that never waits (a pass dropped after its last exchange) must not grow `open` without bound
    auto mine_t = std::make_shared<TicketState>();
    const int64_t tno = c->next_ticket++;
    c->open[tno] = mine_t;
    if (ticket) *ticket = tno;
    for (int side = 0; side < 2; side++)
    {
        const int upper = side == 0 ? c->rank - 1 : c->rank;
        if (upper < 0 || upper + 1 >= c->world) continue;
        Boundary& b = g.boundaries[upper];
        const int me = side == 1 ? 0 : 1, other = 1 - me;   // index inside the boundary: 0 = upper rank
        int us0, us1, ls0, ls1;
        plan(bounds, rows, upper, us0, us1, ls0, ls1);
        Post mine;
        mine.images.assign(images, images + n_images);
        mine.s0 = me == 0 ? us0 : ls0; mine.s1 = me == 0 ? us1 : ls1;
        mine.ticket = mine_t;
        hipError_t e = hipEventCreateWithFlags(&mine.ready, hipEventDisableTiming);
        if (e == hipSuccess && (e = hipEventRecord(mine.ready, cs)) != hipSuccess) (void)hipEventDestroy(mine.ready);
        if (e != hipSuccess)
        {
            // nothing of this side was queued; a first side that was stays posted (its partner will complete it) and keeps the ticket
            if (mine_t->open == 0) c->open.erase(tno);
            hr::set_last_error(std::string("hr_comm_exchange_rows (loopback post): ") + hipGetErrorString(e));
            return HR_ERR_HIP;
        }
        mine_t->open++;
        b.q[me].push_back(mine);
        while (!b.q[0].empty() && !b.q[1].empty())
        {
            // a pair is complete: whoever completes it moves the rows of BOTH directions on its own compute stream once the peer's
            // rows are final, and leaves a done-event the peer's wait waits for.  Any failure below puts nothing back: both posts are
            // closed with their events destroyed, so the peer sees "met" and the error is reported here, where it happened
            Post a = b.q[me].front(), t = b.q[other].front();
            b.q[0].pop_front(); b.q[1].pop_front();
            hipEvent_t done_a = nullptr, done_t = nullptr;
            auto close_pair = [&](hr_status st, const std::string& msg) {
                (void)hipEventDestroy(a.ready); (void)hipEventDestroy(t.ready);
                if (st != HR_OK) { if (done_a) (void)hipEventDestroy(done_a); if (done_t) (void)hipEventDestroy(done_t); hr::set_last_error(msg); }
                else { a.ticket->done.push_back(done_a); t.ticket->done.push_back(done_t); }
                a.ticket->open--; t.ticket->open--;
                g_cv.notify_all();
                return st;
            };
            if (a.images.size() != t.images.size()) return close_pair(HR_ERR_INVALID_ARG, "hr_comm_exchange_rows: ranks passed different image lists");
            hipError_t he = hipStreamWaitEvent(cs, t.ready, 0);
            if (he == hipSuccess) he = hipStreamWaitEvent(cs, a.ready, 0);
            for (size_t i = 0; i < a.images.size() && he == hipSuccess; i++)
            {
                const int64_t pitch = a.images[i].row_pitch_bytes;
                // their rows -> my copy, my rows -> their copy (absolute rows: the same offsets in both images)
                he = hipMemcpyAsync((char*)a.images[i].data + (int64_t)t.s0 * pitch, (char*)t.images[i].data + (int64_t)t.s0 * pitch,
                                    (size_t)((int64_t)(t.s1 - t.s0) * pitch), hipMemcpyDeviceToDevice, cs);
                if (he == hipSuccess)
                    he = hipMemcpyAsync((char*)t.images[i].data + (int64_t)a.s0 * pitch, (char*)a.images[i].data + (int64_t)a.s0 * pitch,
                                        (size_t)((int64_t)(a.s1 - a.s0) * pitch), hipMemcpyDeviceToDevice, cs);
            }
            if (he == hipSuccess) he = hipEventCreateWithFlags(&done_a, hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventCreateWithFlags(&done_t, hipEventDisableTiming);
            if (he == hipSuccess) he = hipEventRecord(done_a, cs);
            if (he == hipSuccess) he = hipEventRecord(done_t, cs);
            if (he != hipSuccess) return close_pair(HR_ERR_HIP, std::string("hr_comm_exchange_rows (loopback copy): ") + hipGetErrorString(he));
            (void)close_pair(HR_OK, "");
        }
    }
    return HR_OK;
}

// everything posted on `c` up to and including `ticket` is complete as far as `compute_stream` is concerned
hr_status hr_comm_wait_ticket(hr_comm* c, hr_comm_ticket ticket, void* compute_stream_)
{
    if (!c) return HR_ERR_INVALID_ARG;
    hipStream_t cs = (hipStream_t)compute_stream_;
    if (c->world == 1 || ticket <= 0) return HR_OK;
    CK_HIP(hipSetDevice(c->device));
    if (ticket >= c->next_ticket) ticket = c->next_ticket - 1;
    if (ticket <= 0) return HR_OK;
    if (!c->loopback)
    {
        // comm_stream runs in order: a ticket whose ring slot has been reused is covered by the newest event
        const int64_t t = (c->next_ticket - 1) - ticket >= kTicketRing ? c->next_ticket - 1 : ticket;
        CK_HIP(hipStreamWaitEvent(cs, c->ring[t % kTicketRing], 0));
        return HR_OK;
    }
    // ranks on their own host threads: block until every post up to `ticket` has met its partner (10 s: a host that waits for a
    // collective its neighbour never posts gets HR_ERR_TIMEOUT instead of a dead lock)
    std::unique_lock<std::mutex> lk(g_mu);
    auto unmatched = [&] {
        for (auto& kv : c->open)
            if (kv.first <= ticket && kv.second->open > 0) return true;
        return false;
    };
    if (!g_cv.wait_for(lk, std::chrono::seconds(10), [&] { return !unmatched(); }))
    {
        hr::set_last_error("hr_comm_wait (loopback): a neighbour has not posted its side of an exchange / all-gather within 10 s");
        return HR_ERR_TIMEOUT;
    }
    // One communicator serves several compute streams (the forked tiled frame waits for its shadows / AO / DDGI tickets on three
    // of them): a wait only ORDERS `cs` behind the done-events of the collectives up to `ticket`, it does not consume them — another
    // stream may still have to wait for an older ticket.  Entries are retired once they are kTicketRing tickets old (prune()).
    hr_status st = HR_OK;
    for (auto it = c->open.begin(); it != c->open.end() && it->first <= ticket; ++it)
        for (hipEvent_t e : it->second->done)
            if (st == HR_OK && hipStreamWaitEvent(cs, e, 0) != hipSuccess) { hr::set_last_error("hr_comm_wait: hipStreamWaitEvent failed"); st = HR_ERR_HIP; }
    prune(c);
    return st;
}

hr_status hr_comm_wait(hr_comm* c, void* compute_stream)
{
    if (!c) return HR_ERR_INVALID_ARG;
    return hr_comm_wait_ticket(c, c->next_ticket - 1, compute_stream);
}

hr_status hr_comm_allgather_rows(hr_comm* c, hr_comm_image image, const int32_t* rb, void* compute_stream_, hr_comm_ticket* ticket)
{
    if (ticket) *ticket = 0;
    if (!c || !image.data || !rb) { hr::set_last_error("hr_comm_allgather_rows: invalid argument"); return HR_ERR_INVALID_ARG; }
    if (c->world == 1) return HR_OK;
    hipStream_t cs = (hipStream_t)compute_stream_;
    CK_HIP(hipSetDevice(c->device));
    const int64_t pitch = image.row_pitch_bytes;
    if (!c->loopback)
    {
        CK_HIP(hipEventRecord(c->ev_compute, cs));
        CK_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_compute, 0));
        CK_NCCL(rccl().GroupStart());
        int err = 0;
        const size_t mine = (size_t)((int64_t)(rb[c->rank + 1] - rb[c->rank]) * pitch);
        for (int p = 0; p < c->world && !err; p++)
        {
            if (p == c->rank) continue;
            if (mine) err = rccl().Send((char*)image.data + (int64_t)rb[c->rank] * pitch, mine, kNcclInt8, p, c->nccl, c->comm_stream);
            const size_t theirs = (size_t)((int64_t)(rb[p + 1] - rb[p]) * pitch);
            if (theirs && !err) err = rccl().Recv((char*)image.data + (int64_t)rb[p] * pitch, theirs, kNcclInt8, p, c->nccl, c->comm_stream);
        }
        const int end = rccl().GroupEnd();
        if (err || end) { hr::set_last_error(std::string("hr_comm_allgather_rows: ") + (rccl().GetErrorString ? rccl().GetErrorString(err ? err : end) : "nccl error")); return HR_ERR_COMM; }
        hr_comm_ticket t = 0;
        hr_status      st = stamp(c, &t);
        if (st != HR_OK) return st;
        if (ticket) *ticket = t;
        CK_HIP(hipStreamWaitEvent(cs, c->ring[t % kTicketRing], 0));   // the readers (probe-grid sample, reflections) follow on compute_stream
        return HR_OK;
    }
    // loopback: the k-th gather of every rank meets; the rank completing a round enqueues all copies on its own stream and leaves
    // done-events for the others (hr_comm_wait_ticket before reading the gathered rows)
    std::lock_guard<std::mutex> lk(g_mu);
    Group& g = g_groups[c->name];
    GatherPost mine;
    mine.image = image;
    CK_HIP(hipEventCreateWithFlags(&mine.ready, hipEventDisableTiming));
    {
        const hipError_t e = hipEventRecord(mine.ready, cs);
        if (e != hipSuccess) { (void)hipEventDestroy(mine.ready); hr::set_last_error(std::string("hr_comm_allgather_rows (loopback post): ") + hipGetErrorString(e)); return HR_ERR_HIP; }
    }
    mine.ticket = std::make_shared<TicketState>();
    mine.ticket->open = 1;
    const int64_t tno = c->next_ticket++;
    c->open[tno] = mine.ticket;   // registered only now that the post cannot fail any more
    if (ticket) *ticket = tno;
    g.gather[c->rank].push_back(mine);
    for (;;)
    {
        bool all = true;
        for (auto& dq : g.gather) all = all && !dq.empty();
        if (!all) break;
        std::vector<GatherPost> round;
        for (auto& dq : g.gather) { round.push_back(dq.front()); dq.pop_front(); }
        hipError_t he = hipSuccess;
        for (GatherPost& q : round) if (he == hipSuccess) he = hipStreamWaitEvent(cs, q.ready, 0);
        for (int dst = 0; dst < g.world; dst++)
            for (int src = 0; src < g.world; src++)
            {
                const size_t bytes = (size_t)((int64_t)(rb[src + 1] - rb[src]) * pitch);
                if (src == dst || !bytes || he != hipSuccess) continue;
                he = hipMemcpyAsync((char*)round[dst].image.data + (int64_t)rb[src] * pitch, (char*)round[src].image.data + (int64_t)rb[src] * pitch, bytes,
                                    hipMemcpyDeviceToDevice, cs);
            }
        for (int p = 0; p < g.world; p++)
        {
            hipEvent_t done = nullptr;
            if (he == hipSuccess) he = hipEventCreateWithFlags(&done, hipEventDisableTiming);
            if (he == hipSuccess && done) he = hipEventRecord(done, cs);
            if (he == hipSuccess && done) round[p].ticket->done.push_back(done);
            else if (done) (void)hipEventDestroy(done);
            (void)hipEventDestroy(round[p].ready);
            round[p].ticket->open--;
        }
        g_cv.notify_all();
        if (he != hipSuccess) { hr::set_last_error(std::string("hr_comm_allgather_rows (loopback copy): ") + hipGetErrorString(he)); return HR_ERR_HIP; }
    }
    return HR_OK;
}

// ---- per-pass conveniences ---------------------------------------------------------------------------------------------
static hr_comm_image as_image(const hr_image_view& v) { return hr_comm_image { v.data, (int64_t)v.row_pitch_bytes }; }

hr_status hr_shadows_exchange_history(hr_shadows* p, hr_comm* c, const int32_t* bounds, int32_t ping_pong, int32_t rows, void* cs, hr_comm_ticket* ticket)
{
    if (!p) return HR_ERR_INVALID_ARG;
    hr_image_view prev, mom;
    hr_status s;
    if ((s = hr_shadows_image(p, 4, &prev)) != HR_OK) return s;                    // feedback image (a-trous iteration `feedback_iteration`)
    if ((s = hr_shadows_image(p, ping_pong ? 3 : 2, &mom)) != HR_OK) return s;     // moments written this frame
    const hr_comm_image im[2] = { as_image(prev), as_image(mom) };
    return hr_comm_exchange_rows(c, im, 2, bounds, rows, cs, ticket);
}

hr_status hr_ao_exchange_history(hr_ao* p, hr_comm* c, const int32_t* bounds, int32_t ping_pong, int32_t rows, void* cs, hr_comm_ticket* ticket)
{
    if (!p) return HR_ERR_INVALID_ARG;
    hr_image_view ao, len;
    hr_status s;
    if ((s = hr_ao_image(p, ping_pong ? 2 : 1, &ao)) != HR_OK) return s;
    if ((s = hr_ao_image(p, ping_pong ? 4 : 3, &len)) != HR_OK) return s;
    const hr_comm_image im[2] = { as_image(ao), as_image(len) };
    return hr_comm_exchange_rows(c, im, 2, bounds, rows, cs, ticket);
}

hr_status hr_reflections_exchange_history(hr_reflections* p, hr_comm* c, const int32_t* bounds, int32_t ping_pong, int32_t rows, void* cs, hr_comm_ticket* ticket)
{
    if (!p) return HR_ERR_INVALID_ARG;
    hr_image_view prev, mom;
    hr_status s;
    // the colour history of the next frame: the feedback image only with blur_as_input, else THIS frame's temporal output (the reference's
    // default; round 3: the feedback image was exchanged unconditionally, so with the default parameters the apron rows kept locally
    // computed history that degrades by the per-frame vertical motion — found by the 8-band 4K test of tests/test_gpu_comm.py)
    if ((s = hr_reflections_image(p, 10, &prev)) != HR_OK) return s;
    if ((s = hr_reflections_image(p, ping_pong ? 4 : 3, &mom)) != HR_OK) return s;
    const hr_comm_image im[2] = { as_image(prev), as_image(mom) };
    return hr_comm_exchange_rows(c, im, 2, bounds, rows, cs, ticket);
}

hr_status hr_ddgi_allgather_atlases(hr_ddgi* p, hr_comm* c, void* cs, hr_comm_ticket* ticket)
{
    if (ticket) *ticket = 0;
    if (!p || !c) return HR_ERR_INVALID_ARG;
    hr_ddgi_uniforms u;
    hr_image_view    irr, dep;
    hr_status        s;
    if ((s = hr_ddgi_get_uniforms(p, &u)) != HR_OK) return s;
    if ((s = hr_ddgi_current_write(p, &irr, &dep)) != HR_OK) return s;
    const int cz = u.probe_counts[2], world = hr_comm_world(c);
    if (cz < world) { hr::set_last_error("hr_ddgi_allgather_atlases: fewer probe z-slabs than ranks"); return HR_ERR_INVALID_ARG; }
    // probes are laid out x + y * cx along the atlas x axis and z along y, each (side + 2) texels, inside a one-texel frame
    // (ddgi.cpp:197-201): the probes of z-slabs [z0, z1) are the contiguous atlas rows 1 + z0 (side + 2) .. 1 + z1 (side + 2)
    std::vector<int32_t> rb(world + 1);
    for (int pass = 0; pass < 2; pass++)
    {
        const int side = pass == 0 ? u.irradiance_probe_side_length : u.depth_probe_side_length;
        for (int r = 0; r <= world; r++) rb[r] = 1 + (int)(((long long)cz * r) / world) * (side + 2);
        if ((s = hr_comm_allgather_rows(c, as_image(pass == 0 ? irr : dep), rb.data(), cs, ticket)) != HR_OK) return s;   // the second ticket covers both
    }
    return HR_OK;
}

} // extern "C"
