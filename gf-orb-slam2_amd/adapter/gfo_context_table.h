// gfo_context_table.h -- which gfo context belongs to which ORBextractor object (adapter/ORBextractor_gfo.cc).
//
// include/ORBextractor.h of the reference cannot carry a new member and its destructor is inline and empty (:84), so the
// adapter never sees an extractor die.  The table therefore keeps, per object ADDRESS, the constructor arguments and a
// context created on first use, under these rules:
//
//   * an extractor that is in use is never evicted, however many there are: a context is only reclaimed when its owner
//     has not called for `idle_limit()` table lookups (64 x the number of live contexts, at least 4096) -- a rig of K
//     cameras calling round-robin touches every context once per ~K lookups, three orders of magnitude inside the limit;
//   * a context is PINNED while a call is inside the library with it (ContextTable::Use, the only way the adapters obtain a
//     context): reclaim() -- run from whichever thread has just created a context -- skips pinned entries, so neither the LRU
//     ceiling nor the idle rule can destroy a context under a thread that is using it (get() used to hand out a raw
//     pointer and drop the lock); a constructor at the address of a pinned entry defers the old context's destruction to the
//     moment its last user leaves;
//   * a constructor at an address the table already knows means the old object is gone (Tracking::updateORBExtractor,
//     src/Tracking.cc:298-320, deletes and re-creates both extractors; glibc hands `new` the block `delete` just freed):
//     its context is destroyed at once and the new object gets a NEW context -- nothing cached "per address"
//     (uploaded vocabularies, matchers_gfo.cc) may survive that, which is why residency is asked of the context itself
//     (gfo_ctx_id / gfo_vocabulary_nodes) and never remembered by pointer;
//   * when the allocator does NOT reuse the address, the dead object's context simply goes idle and is reclaimed by the
//     first rule; GFO_MAX_CONTEXTS (default 64) is only a ceiling against a runaway, enforced least-recently-used.
//
//   * SEVERAL GPUs (GFO_DEVICES=0,1,2,...; default: the one device GFO_DEVICE names, or 0): an extractor is placed when it is
//     declared, on the listed device that carries the fewest extractors (a re-declared address keeps its device); the two
//     extractors of a stereo rig must share a device -- the library pairs them into one submission (gfo_ctx_pair) and every matcher
//     call of a frame runs in its LEFT extractor's context -- so the stereo member calls colocate(right, left) before it pins
//     them: a rig found on two devices moves to the emptier of the two (the extractor that moves gives up its context and gets a
//     new one there at its next call; never one with a call in flight).  One move per rig at most, on its first frame, after
//     which K rigs sit on K devices; K camera threads of one process then spread over the node's GPUs, which the reference's
//     `System` per camera cannot say by itself.
//
// No OpenCV in this header: tests/test_host_logic.py builds it against a counting stand-in of the four C entry points
// it calls and drives exactly the scenarios above on the CPU.
#pragma once

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "gfo.h"

namespace gfo_adapter
{

class ContextTable
{
public:
    // called on every context the table creates (the adapter switches the frame combiner on there)
    void (*on_create)(gfo_ctx*) = NULL;

    struct Entry {
        gfo_params prm;
        gfo_ctx* ctx;
        unsigned long stamp;
        int in_use;            // calls inside the library with `ctx` (or `retired`) right now
        gfo_ctx* retired;      // the context of a dead object at this address that still had a user when the new object was declared
        int slot;              // index into devices(): where this extractor's context lives (or will be created)
        bool fresh;            // declared, no context created yet (counts as load of its slot)
    };

    // GFO_DEVICES=0,1,...: the devices extractors are spread over; default: one entry, GFO_DEVICE (or 0).  A device may be listed
    // more than once (two slots on one GPU; the tests do that to exercise placement and moves on a one-GPU box).
    // (read from the environment at the table's first use; set_devices() replaces the list -- tests, or an application that knows
    //  its node -- before any extractor exists)
    void set_devices(const std::vector<int>& d)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (!d.empty()) devs_ = d;
    }
    int device_count()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return (int)devices().size();
    }

    // A context for the duration of one call: pins the entry so that reclaim() on another thread cannot destroy it.
    //   ContextTable::Use u(table, this);  if (gfo_ctx* c = u.ctx()) gfo_extract(c, ...);
    class Use
    {
    public:
        Use(ContextTable& t, const void* key) : t_(t), key_(key), ctx_(t.acquire(key)) {}
        ~Use() { if (ctx_) t_.release(key_, ctx_); }
        gfo_ctx* ctx() const { return ctx_; }
    private:
        Use(const Use&);
        Use& operator=(const Use&);
        ContextTable& t_;
        const void* key_;
        gfo_ctx* ctx_;
    };

    // the constructor of an extractor: remembers its arguments; a context the address still owns belongs to a dead object
    void declare(const void* key, const gfo_params& prm)
    {
        std::lock_guard<std::mutex> lk(mu_);
        std::map<const void*, Entry>::iterator it = tab_.find(key);
        Entry e;
        e.prm = prm;
        e.ctx = NULL;
        e.stamp = ++clock_;
        e.in_use = 0;
        e.retired = NULL;
        e.fresh = true;
        e.slot = it != tab_.end() ? it->second.slot : least_loaded_slot_locked();   // a re-created extractor stays where it was
        if (it != tab_.end()) {
            Entry& old = it->second;
            if (old.ctx && old.in_use > 0 && !old.retired) {
                // somebody is still inside a call with the dead object's context (the caller's bug, but not ours to crash on):
                // it is destroyed when that call returns
                e.retired = old.ctx;
                e.in_use = old.in_use;
            } else if (old.ctx) {
                gfo_ctx_destroy(old.ctx);
                destroyed_++;
            }
            if (old.retired) {       // (a second re-declaration while the first dead context is still in use: keep waiting for it)
                e.retired = old.retired;
                e.in_use = old.in_use;
            }
        }
        tab_[key] = e;
    }

    // pin + get: the context of a live extractor for the duration of a call (see Use); release() with what acquire() returned
    gfo_ctx* acquire(const void* key)
    {
        std::lock_guard<std::mutex> lk(mu_);
        gfo_ctx* c = get_locked(key);
        if (c) tab_[key].in_use++;
        return c;
    }

    void release(const void* key, gfo_ctx* c)
    {
        std::lock_guard<std::mutex> lk(mu_);
        std::map<const void*, Entry>::iterator it = tab_.find(key);
        if (it == tab_.end()) return;
        Entry& e = it->second;
        if (e.in_use > 0) e.in_use--;
        if (e.in_use == 0 && e.retired) {      // the last user of a dead object's context has left
            if (e.retired != e.ctx) {
                gfo_ctx_destroy(e.retired);
                destroyed_++;
            }
            e.retired = NULL;
        }
        (void)c;
    }

    // The two extractors of a stereo rig must live on one device.  When they do not, both go to whichever of their two slots carries
    // fewer OTHER extractors (ties: the leader's) -- extractors are declared before anybody knows which two form a rig, and moving
    // the pair to the emptier side is what leaves K rigs spread over K devices instead of piled onto the leaders' -- and the one
    // that moves gives up its context: the next acquire() creates one on the new device.  Returns true when something moved.
    // An extractor with a call in flight is never moved under its user (the other one moves instead, or the caller asks again
    // with the next frame).
    bool colocate(const void* follower, const void* leader)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (devices().size() < 2) return false;
        std::map<const void*, Entry>::iterator f = tab_.find(follower), l = tab_.find(leader);
        if (f == tab_.end() || l == tab_.end() || f->second.slot == l->second.slot) return false;
        int others_f = 0, others_l = 0;
        for (std::map<const void*, Entry>::const_iterator it = tab_.begin(); it != tab_.end(); ++it) {
            if (it == f || it == l || !(it->second.ctx || it->second.fresh)) continue;
            others_f += it->second.slot == f->second.slot;
            others_l += it->second.slot == l->second.slot;
        }
        std::map<const void*, Entry>::iterator mover = others_f < others_l ? l : f, stays = others_f < others_l ? f : l;
        if (mover->second.in_use > 0) std::swap(mover, stays);
        if (mover->second.in_use > 0) return false;
        if (mover->second.ctx) {
            gfo_ctx_destroy(mover->second.ctx);
            mover->second.ctx = NULL;
            destroyed_++;
        }
        mover->second.slot = stays->second.slot;
        moved_++;
        return true;
    }

    // the device (its HIP ordinal) an extractor is placed on; -1 for an address the table does not know
    int device_of(const void* key)
    {
        std::lock_guard<std::mutex> lk(mu_);
        std::map<const void*, Entry>::iterator it = tab_.find(key);
        return it == tab_.end() ? -1 : devices()[it->second.slot];
    }
    int slot_of(const void* key)
    {
        std::lock_guard<std::mutex> lk(mu_);
        std::map<const void*, Entry>::iterator it = tab_.find(key);
        return it == tab_.end() ? -1 : it->second.slot;
    }
    unsigned long moved() const { return moved_; }

    // the context of a live extractor, created on first use; NULL (and a message) when the device refuses.  UNPINNED: for
    // single-threaded callers and the tests; the adapters go through Use.
    gfo_ctx* get(const void* key)
    {
        std::lock_guard<std::mutex> lk(mu_);
        return get_locked(key);
    }

    void destroy_all()
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (std::map<const void*, Entry>::iterator it = tab_.begin(); it != tab_.end(); ++it) {
            if (it->second.ctx) {
                gfo_ctx_destroy(it->second.ctx);
                it->second.ctx = NULL;
                destroyed_++;
            }
            if (it->second.retired) {
                gfo_ctx_destroy(it->second.retired);
                it->second.retired = NULL;
                destroyed_++;
            }
        }
    }

    int alive()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return alive_locked();
    }
    unsigned long created() const { return created_; }
    unsigned long destroyed() const { return destroyed_; }

    static int max_contexts()
    {
        static const int n = getenv("GFO_MAX_CONTEXTS") ? atoi(getenv("GFO_MAX_CONTEXTS")) : 64;
        return n < 2 ? 2 : n;
    }

private:
    gfo_ctx* get_locked(const void* key)
    {
        std::map<const void*, Entry>::iterator it = tab_.find(key);
        if (it == tab_.end()) return NULL;   // not constructed through the adapter
        Entry& e = it->second;
        e.stamp = ++clock_;
        if (!e.ctx) {
            const int dev = devices()[e.slot];
            if (gfo_ctx_create(&e.prm, dev, &e.ctx) != GFO_OK) {
                fprintf(stderr, "[gfo] ORBextractor: %s\n", gfo_last_error(NULL));
                e.ctx = NULL;
                return NULL;
            }
            created_++;
            e.fresh = false;
            if (on_create) on_create(e.ctx);
            reclaim(key);
        }
        return e.ctx;
    }

    // the caller holds mu_
    const std::vector<int>& devices()
    {
        if (devs_.empty()) devs_ = parse_devices();
        return devs_;
    }

    static std::vector<int> parse_devices()
    {
        std::vector<int> v;
        if (const char* s = getenv("GFO_DEVICES")) {
            for (const char* p = s; *p;) {
                char* end = NULL;
                const long d = strtol(p, &end, 10);
                if (end == p) break;
                if (d >= 0 && d < 1024) v.push_back((int)d);
                p = end;
                while (*p == ',' || *p == ' ') p++;
            }
        }
        if (v.empty()) v.push_back(getenv("GFO_DEVICE") ? atoi(getenv("GFO_DEVICE")) : 0);
        return v;
    }

    // the slot with the fewest extractors on it: live contexts and declared-but-not-yet-used entries count, the entries of
    // extractors that went idle and lost their context do not; ties go to the slot listed first.  The caller holds mu_.
    int least_loaded_slot_locked()
    {
        const size_t n = devices().size();
        if (n < 2) return 0;
        std::vector<int> load(n, 0);
        for (std::map<const void*, Entry>::const_iterator it = tab_.begin(); it != tab_.end(); ++it)
            if (it->second.ctx || it->second.fresh) load[it->second.slot]++;
        size_t best = 0;
        for (size_t k = 1; k < n; k++)
            if (load[k] < load[best]) best = k;
        return (int)best;
    }

    int alive_locked() const
    {
        int n = 0;
        for (std::map<const void*, Entry>::const_iterator it = tab_.begin(); it != tab_.end(); ++it) n += it->second.ctx != NULL;
        return n;
    }

    // runs only when a context has just been created (never on the per-frame path); the caller holds mu_
    void reclaim(const void* keep)
    {
        for (;;) {
            const int alive = alive_locked();
            const unsigned long idle_limit = 64ul * (unsigned long)alive > 4096ul ? 64ul * (unsigned long)alive : 4096ul;
            std::map<const void*, Entry>::iterator oldest = tab_.end();
            for (std::map<const void*, Entry>::iterator it = tab_.begin(); it != tab_.end(); ++it) {
                if (!it->second.ctx || it->first == keep || it->second.in_use > 0) continue;   // never the caller's own, never one in use
                if (oldest == tab_.end() || it->second.stamp < oldest->second.stamp) oldest = it;
            }
            if (oldest == tab_.end()) return;
            const bool idle = clock_ - oldest->second.stamp > idle_limit;
            if (!idle && alive <= max_contexts()) return;
            gfo_ctx_destroy(oldest->second.ctx);   // an owner that does still exist gets a fresh context on its next call
            oldest->second.ctx = NULL;
            destroyed_++;                          // (the entry itself stays: an extractor that was merely idle --
                                                   //  mpIniORBextractor between two initialisations -- must still be known)
        }
    }

    std::mutex mu_;
    std::map<const void*, Entry> tab_;
    std::vector<int> devs_;
    unsigned long clock_ = 0, created_ = 0, destroyed_ = 0, moved_ = 0;
};

}  // namespace gfo_adapter
