// gfo_context_table.h -- which gfo context belongs to which ORBextractor object (adapter/ORBextractor_gfo.cc).
//
// include/ORBextractor.h of the reference cannot carry a new member and its destructor is inline and empty (:84), so the
// adapter never sees an extractor die.  The table therefore keeps, per object ADDRESS, the constructor arguments and a
// context created on first use, under these rules:
//
//   * an extractor that is in use is never evicted, however many there are: a context is only reclaimed when its owner
//     has not called for `idle_limit()` table lookups (64 x the number of live contexts, at least 4096) -- a rig of K
//     cameras calling round-robin touches every context once per ~K lookups, three orders of magnitude inside the limit;
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
    };

    // the constructor of an extractor: remembers its arguments; a context the address still owns belongs to a dead object
    void declare(const void* key, const gfo_params& prm)
    {
        std::lock_guard<std::mutex> lk(mu_);
        std::map<const void*, Entry>::iterator it = tab_.find(key);
        if (it != tab_.end() && it->second.ctx) {
            gfo_ctx_destroy(it->second.ctx);
            destroyed_++;
        }
        Entry e;
        e.prm = prm;
        e.ctx = NULL;
        e.stamp = ++clock_;
        tab_[key] = e;
    }

    // the context of a live extractor, created on first use; NULL (and a message) when the device refuses
    gfo_ctx* get(const void* key)
    {
        std::lock_guard<std::mutex> lk(mu_);
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

    void destroy_all()
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (std::map<const void*, Entry>::iterator it = tab_.begin(); it != tab_.end(); ++it)
            if (it->second.ctx) {
                gfo_ctx_destroy(it->second.ctx);
                it->second.ctx = NULL;
                destroyed_++;
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
                if (!it->second.ctx || it->first == keep) continue;
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
