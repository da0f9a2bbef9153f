// s2m_wait.h -- every wait of the host: on a word the device publishes, on a stream, on an event, on the handle's side thread.
//
// The reference's loop never blocks on its map (laserMapping.cpp:726-731; a search only waits for the rebuild thread's
// short critical sections, ikd_Tree.cpp:436-449).  This engine's caller waits for the device several times per frame, and a
// wait that cannot end must not hold the node for ever: every wait has ONE deadline (s2m_config.wait_timeout_ms) and ONE policy
// (s2m_config.wait_policy) and reports what it was waiting for when the deadline passes.  No wait in the product calls a
// blocking runtime entry (hipStreamSynchronize, hipEventSynchronize, a condition variable without a deadline).
//
// Policies (the pool's "busy host" boxes are what a robot's CPU looks like: a thread that spins through a whole kernel is a
// thread the scheduler takes away at the wrong moment):
//   spin   the calling thread polls with `pause` until the deadline -- lowest latency, one core busy (the default; what the
//          benchmark's headline is quoted with)
//   yield  polls for wait_spin_us, then sched_yield() between polls -- the core is shared with whoever is runnable
//   sleep  polls for wait_spin_us, then nanosleep between polls (50 us, 200 us after 5 ms) -- the core is free while the
//          device works; costs up to one sleep quantum per wait that outlasts the spin phase
// Fault injection (tests only, S2M_TEST_STALL=worker|mail|reduce[:after]): one hand-back is withheld so that the deadline
// path is exercised on a healthy device.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <time.h>

#include <atomic>
#include <chrono>
#include <cstdint>

namespace s2m {

enum { kWaitSpin = 0, kWaitYield = 1, kWaitSleep = 2 };
enum { kStallNone = 0, kStallWorker = 1, kStallMail = 2, kStallReduce = 3 };
// the runtime's own code for "the device did not answer in time": what a wait that expired reports to its caller
constexpr hipError_t kWaitTimedOut = hipErrorLaunchTimeOut;

struct WaitCtl {
    int policy = kWaitSpin;
    int64_t timeout_us = 10 * 1000 * 1000;
    int64_t spin_us = 40;
    // fault injection: the stall_after-th hand-back of kind stall_kind (and every later one) is withheld
    int stall_kind = kStallNone;
    std::atomic<long> stall_after{0};
    // the wait that expired (first one wins: later ones are consequences)
    std::atomic<const char *> expired{nullptr};
    std::atomic<int64_t> waited_us{0};
    // diagnostics (S2M_HOSTTIME; the stall report): waits begun, waits that outlasted the spin phase, time spent waiting
    std::atomic<int64_t> n_waits{0}, n_slow{0}, total_ns{0};
    bool hosttime = false;

    // true when the hook says "withhold this one"
    bool withhold(int kind)
    {
        if (stall_kind != kind) return false;
        return stall_after.fetch_sub(1, std::memory_order_relaxed) <= 0;
    }
};

inline WaitCtl *default_wait()
{
    static WaitCtl w;
    return &w;
}
// the wait control of the handle whose entry point the calling thread is inside (set on entry, s2m_engine_internal.h; the
// handle's side thread sets it once): what the layers without a handle in reach -- mailboxes, map code -- wait with
inline thread_local WaitCtl *tl_wait = nullptr;
inline WaitCtl *cur_wait() { return tl_wait ? tl_wait : default_wait(); }

inline int64_t wait_elapsed_us(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
}

// Polls seen() under the policy until it is true (-> true) or `timeout_us` has passed (-> false); *waited = how long.
template <class Seen>
inline bool wait_core(WaitCtl *wc, Seen &&seen, int64_t timeout_us, int64_t *waited)
{
    for (int i = 0; i < 32; ++i) {   // most hand-backs are here within a microsecond or two: no clock read
        if (seen()) return true;
        __builtin_ia32_pause();
    }
    const auto t0 = std::chrono::steady_clock::now();
    wc->n_waits.fetch_add(1, std::memory_order_relaxed);
    const int64_t spin_for = wc->policy == kWaitSpin ? timeout_us : wc->spin_us;
    int64_t us = 0;
    bool ok = false;
    for (;;) {
        for (int i = 0; i < 64 && !ok; ++i) {
            ok = seen();
            if (!ok) __builtin_ia32_pause();
        }
        if (ok) break;
        us = wait_elapsed_us(t0);
        if (us > spin_for) break;
    }
    if (!ok && wc->policy != kWaitSpin) {
        wc->n_slow.fetch_add(1, std::memory_order_relaxed);
        for (;;) {
            ok = seen();
            if (ok) break;
            us = wait_elapsed_us(t0);
            if (us > timeout_us) break;
            if (wc->policy == kWaitYield) {
                sched_yield();
            } else {
                struct timespec ts = {0, us > 5000 ? 200000L : 50000L};
                nanosleep(&ts, nullptr);
            }
        }
    }
    if (wc->hosttime)
        wc->total_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(),
                               std::memory_order_relaxed);
    if (waited) *waited += us;
    return ok;
}

inline void wait_expired(WaitCtl *wc, const char *what, int64_t us)
{
    const char *none = nullptr;
    if (wc->expired.compare_exchange_strong(none, what)) wc->waited_us.store(us, std::memory_order_relaxed);
}

// ... with the handle's deadline; on expiry wc->expired names the wait
template <class Seen>
inline bool wait_until(WaitCtl *wc, Seen &&seen, const char *what)
{
    if (!wc) wc = cur_wait();
    int64_t us = 0;
    if (wait_core(wc, seen, wc->timeout_us, &us)) return true;
    wait_expired(wc, what, us);
    return false;
}

// everything enqueued on the stream has finished (hipStreamSynchronize with the handle's policy and deadline)
inline hipError_t wait_stream(WaitCtl *wc, hipStream_t st, const char *what)
{
    hipError_t q = hipSuccess;
    if (!wait_until(wc, [&] { q = hipStreamQuery(st); return q != hipErrorNotReady; }, what)) return kWaitTimedOut;
    return q;
}

inline hipError_t wait_event(WaitCtl *wc, hipEvent_t ev, const char *what)
{
    hipError_t q = hipSuccess;
    if (!wait_until(wc, [&] { q = hipEventQuery(ev); return q != hipErrorNotReady; }, what)) return kWaitTimedOut;
    return q;
}

// A word in pinned host memory that a kernel on `st` raises to `seq` (system-scope release) once what it guards is written.
// Ten times a second the stream is asked what became of the kernel: a runtime error is reported as such at once, and so is a
// stream that has gone idle without the word ever being raised (a hand-back that was lost will not arrive later either).
template <class Word>
inline hipError_t wait_word(WaitCtl *wc, const volatile Word *flag, Word seq, hipStream_t st, const char *what)
{
    if (!wc) wc = cur_wait();
    auto seen = [&] { return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq; };
    int64_t us = 0;
    for (int64_t left = wc->timeout_us; left > 0;) {
        const int64_t slice = left < 100000 ? left : 100000;
        if (wait_core(wc, seen, slice, &us)) return hipSuccess;
        left -= slice;
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) {
            if (seen()) return hipSuccess;
            break;
        }
        if (q != hipErrorNotReady) return q;
    }
    wait_expired(wc, what, us);
    return kWaitTimedOut;
}

// The polling loops that serve several handles at once (s2m_iterated_update_batch): progress() when a round found a block,
// idle() when it found none -- one idle step under the policy; true once the deadline has passed without any progress.
struct IdleWait {
    WaitCtl *wc;
    std::chrono::steady_clock::time_point t0;
    long spins = 0;
    int64_t us = 0;
    explicit IdleWait(WaitCtl *w) : wc(w ? w : cur_wait()) {}
    void progress() { spins = 0; }
    bool idle()
    {
        __builtin_ia32_pause();
        if (++spins < 64) return false;
        if (spins == 64) { t0 = std::chrono::steady_clock::now(); wc->n_waits.fetch_add(1, std::memory_order_relaxed); return false; }
        if (wc->policy == kWaitSpin && (spins & 63) != 0) return false;
        us = wait_elapsed_us(t0);
        if (us > wc->timeout_us) return true;
        if (wc->policy != kWaitSpin && us > wc->spin_us) {
            if (wc->policy == kWaitYield) {
                sched_yield();
            } else {
                struct timespec ts = {0, us > 5000 ? 200000L : 50000L};
                nanosleep(&ts, nullptr);
            }
        }
        return false;
    }
};

}  // namespace s2m
