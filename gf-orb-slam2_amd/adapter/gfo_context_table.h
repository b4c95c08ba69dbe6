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
// No OpenCV in this header: tests/test_host_logic.py builds it against a counting stand-in of the four C entry points
// it calls and drives exactly the scenarios above on the CPU.
#pragma once

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>

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
    };

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
            int dev = 0;
            if (const char* d = getenv("GFO_DEVICE")) dev = atoi(d);
            if (gfo_ctx_create(&e.prm, dev, &e.ctx) != GFO_OK) {
                fprintf(stderr, "[gfo] ORBextractor: %s\n", gfo_last_error(NULL));
                e.ctx = NULL;
                return NULL;
            }
            created_++;
            if (on_create) on_create(e.ctx);
            reclaim(key);
        }
        return e.ctx;
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
    unsigned long clock_ = 0, created_ = 0, destroyed_ = 0;
};

}  // namespace gfo_adapter
