// context_table_check.cc -- CPU test of adapter/gfo_context_table.h against a counting stand-in of libgfo's
// context entry points (tests/test_host_logic.py builds and runs it; no GPU, no OpenCV).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "gfo_context_table.h"

struct gfo_ctx { uint64_t id; };
static int g_created = 0, g_destroyed = 0;
static std::set<gfo_ctx*> g_live;
extern "C" int gfo_ctx_create(const gfo_params*, int, gfo_ctx** out)
{
    *out = new gfo_ctx{(uint64_t)++g_created};
    g_live.insert(*out);
    return GFO_OK;
}
extern "C" void gfo_ctx_destroy(gfo_ctx* c)
{
    if (!g_live.erase(c)) { printf("FAIL destroy of a dead context\n"); exit(1); }
    g_destroyed++;
    delete c;
}
extern "C" const char* gfo_last_error(const gfo_ctx*) { return "stand-in"; }

#define CHECK(cond) do { if (!(cond)) { printf("FAIL %s:%d %s (created %d destroyed %d)\n", __FILE__, __LINE__, #cond, g_created, g_destroyed); return 1; } } while (0)

int main()
{
    gfo_params prm = {2000, 1.2f, 8, 20, 7, 1};
    // (a) a rig of 16 stereo cameras = 32 extractors, round-robin: every one keeps its context for ever
    {
        gfo_adapter::ContextTable t;
        std::vector<char> objs(32);
        for (size_t i = 0; i < objs.size(); i++) t.declare(&objs[i], prm);
        std::vector<gfo_ctx*> first(objs.size());
        for (size_t i = 0; i < objs.size(); i++) CHECK((first[i] = t.get(&objs[i])) != NULL);
        const int c0 = g_created;
        for (int it = 0; it < 20000; it++)
            for (size_t i = 0; i < objs.size(); i++) CHECK(t.get(&objs[i]) == first[i]);
        CHECK(g_created == c0 && g_destroyed == 0 && t.alive() == 32);
        // (b) Tracking::updateORBExtractor: delete + new at the SAME address -> the old context is retired at once, the new
        //     object gets a context of its own (a different one, whatever the pointer value)
        const uint64_t old_id = first[3]->id;
        t.declare(&objs[3], prm);
        CHECK(g_destroyed == 1);
        gfo_ctx* fresh = t.get(&objs[3]);
        CHECK(fresh != NULL && fresh->id != old_id && g_created == c0 + 1 && t.alive() == 32);
        // (c) ... at a DIFFERENT address: the dead object's context goes idle and is reclaimed once it has been silent for
        //     thousands of lookups, the live ones never are
        std::vector<char> more(4);
        for (size_t i = 0; i < more.size(); i++) { t.declare(&more[i], prm); CHECK(t.get(&more[i]) != NULL); }
        CHECK(t.alive() == 36);
        for (int it = 0; it < 400; it++)            // objs[0..3] are "deleted": nobody calls them again
            for (size_t i = 4; i < objs.size(); i++) CHECK(t.get(&objs[i]) == first[i]);
        for (size_t i = 0; i < more.size(); i++) CHECK(t.get(&more[i]) != NULL);
        char late;
        t.declare(&late, prm);
        CHECK(t.get(&late) != NULL);                // a creation is when the table looks for idle contexts
        CHECK(t.alive() == 36 - 4 + 1);
        for (size_t i = 4; i < objs.size(); i++) CHECK(t.get(&objs[i]) == first[i]);
        // an extractor that was only idle (mpIniORBextractor between two initialisations) is still known and gets a new context
        CHECK(t.get(&objs[0]) != NULL);
        t.destroy_all();
        CHECK(g_live.empty());
    }
    // (d) the ceiling: more live extractors than GFO_MAX_CONTEXTS -> least recently used goes, never the caller's own
    {
        setenv("GFO_MAX_CONTEXTS", "64", 1);
        gfo_adapter::ContextTable t;
        std::vector<char> objs(70);
        for (size_t i = 0; i < objs.size(); i++) { t.declare(&objs[i], prm); CHECK(t.get(&objs[i]) != NULL); }
        CHECK(t.alive() == 64);
        gfo_ctx* mine = t.get(&objs[69]);
        CHECK(mine != NULL && t.get(&objs[69]) == mine);
        t.destroy_all();
        CHECK(g_live.empty());
    }
    // (e) ADVICE r3: a context that is IN USE is never reclaimed -- neither by the ceiling nor by the idle rule -- while another
    //     thread's first call creates contexts; and a constructor at the address of an object that still has a call in flight
    //     defers the old context's destruction to the end of that call
    {
        setenv("GFO_MAX_CONTEXTS", "64", 1);
        gfo_adapter::ContextTable t;
        std::vector<char> objs(80);
        t.declare(&objs[0], prm);
        gfo_ctx* held;
        {
            gfo_adapter::ContextTable::Use use(t, &objs[0]);       // "inside gfo_extract" with objs[0]'s context, the oldest of all
            held = use.ctx();
            CHECK(held != NULL);
            for (size_t i = 1; i < objs.size(); i++) { t.declare(&objs[i], prm); CHECK(t.get(&objs[i]) != NULL); }   // 79 creations, ceiling 64
            for (int it = 0; it < 6000; it++) CHECK(t.get(&objs[79]) != NULL);                                        // ... and 6000 lookups: objs[0] is "idle"
            char late;
            t.declare(&late, prm);
            CHECK(t.get(&late) != NULL);
            CHECK(g_live.count(held) == 1);                          // still alive: it is pinned
            CHECK(t.alive() <= 65);
            // the object is re-created at the same address while the call is still inside
            const int d0 = g_destroyed;
            t.declare(&objs[0], prm);
            CHECK(g_live.count(held) == 1 && g_destroyed == d0);     // not yet
            gfo_ctx* fresh = t.get(&objs[0]);
            CHECK(fresh != NULL && fresh != held);
        }                                                            // the call returns: now the dead object's context goes
        CHECK(g_live.count(held) == 0);
        t.destroy_all();
        CHECK(g_live.empty());
    }
    printf("OK created %d destroyed %d\n", g_created, g_destroyed);
    return 0;
}
