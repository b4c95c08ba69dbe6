// context_table_check.cc -- CPU test of adapter/gfo_context_table.h against a counting stand-in of libgfo's
// context entry points (tests/test_host_logic.py builds and runs it; no GPU, no OpenCV).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "gfo_context_table.h"

struct gfo_ctx { uint64_t id; int device; };
static int g_created = 0, g_destroyed = 0;
static std::set<gfo_ctx*> g_live;
extern "C" int gfo_ctx_create(const gfo_params*, int device, gfo_ctx** out)
{
    *out = new gfo_ctx{(uint64_t)++g_created, device};
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
    // (f) several GPUs (GFO_DEVICES): extractors are placed on the least-loaded device when they are declared; the right extractor
    //     of a stereo rig follows its left one (colocate: what Frame::ComputeStereoMatches_Undistorted asks before it pairs them);
    //     a re-declared address keeps its device; matcher calls use the left extractor's context, hence its device
    {
        gfo_adapter::ContextTable t;
        t.set_devices(std::vector<int>{4, 5, 6, 7});          // HIP ordinals, not indices
        CHECK(t.device_count() == 4);
        // four stereo cameras constructed one after the other, as four `System` objects do (Tracking.cc:247-254: left, then right)
        std::vector<char> L(4), R(4);
        for (int k = 0; k < 4; k++) {
            t.declare(&L[k], prm); CHECK(t.get(&L[k]) != NULL);
            t.declare(&R[k], prm); CHECK(t.get(&R[k]) != NULL);
        }
        int per_dev[8] = {0};
        for (int k = 0; k < 4; k++) { per_dev[t.device_of(&L[k])]++; per_dev[t.device_of(&R[k])]++; }
        CHECK(per_dev[4] == 2 && per_dev[5] == 2 && per_dev[6] == 2 && per_dev[7] == 2);      // spread evenly before any rig is known
        for (int k = 0; k < 4; k++) CHECK(t.get(&L[k])->device == t.device_of(&L[k]) && t.get(&R[k])->device == t.device_of(&R[k]));
        // first frame of every camera: the stereo member co-locates the rig -- on the emptier of the two devices
        int moves = 0;
        for (int k = 0; k < 4; k++) {
            const uint64_t bl = t.get(&L[k])->id, br = t.get(&R[k])->id;
            const bool moved = t.colocate(&R[k], &L[k]);
            moves += moved;
            CHECK(t.device_of(&R[k]) == t.device_of(&L[k]));
            gfo_ctx *l = t.get(&L[k]), *r = t.get(&R[k]);                      // the next operator() calls
            CHECK(l != NULL && r != NULL && l->device == t.device_of(&L[k]) && r->device == l->device);
            CHECK(moved ? (l->id != bl) != (r->id != br) : (l->id == bl && r->id == br));   // exactly one of the two got a new context
            CHECK(!t.colocate(&R[k], &L[k]));                                  // idempotent: nothing to do from the second frame on
        }
        CHECK((unsigned long)moves == t.moved() && moves <= 4);
        // four rigs, four devices: one rig each
        std::set<int> used;
        for (int k = 0; k < 4; k++) used.insert(t.device_of(&L[k]));
        CHECK(used.size() == 4);
        // Tracking::updateORBExtractor re-creates both extractors at their addresses: same devices, no move needed again
        const int dl = t.device_of(&L[1]), dr = t.device_of(&R[1]);
        t.declare(&L[1], prm); t.declare(&R[1], prm);
        CHECK(t.device_of(&L[1]) == dl && t.device_of(&R[1]) == dr && dl == dr);
        CHECK(t.get(&L[1])->device == dl && t.get(&R[1])->device == dr);
        CHECK(!t.colocate(&R[1], &L[1]));
        // a follower with a call in flight is not moved under its user; the next frame moves it
        char L9, R9;
        t.declare(&L9, prm); t.declare(&R9, prm);
        if (t.device_of(&L9) == t.device_of(&R9)) { char pad; t.declare(&pad, prm); t.declare(&R9, prm); }
        CHECK(t.device_of(&L9) != t.device_of(&R9));
        {
            gfo_adapter::ContextTable::Use use(t, &R9), use_l(t, &L9);          // both inside a call: nobody moves
            CHECK(use.ctx() != NULL && use_l.ctx() != NULL);
            CHECK(!t.colocate(&R9, &L9));
            CHECK(g_live.count(use.ctx()) == 1 && g_live.count(use_l.ctx()) == 1);
        }
        {
            gfo_adapter::ContextTable::Use use(t, &R9);                         // one inside a call: the other one is the one that moves
            gfo_ctx* held = use.ctx();
            CHECK(t.colocate(&R9, &L9));
            CHECK(g_live.count(held) == 1 && t.device_of(&L9) == held->device);
        }
        CHECK(!t.colocate(&R9, &L9));
        CHECK(t.device_of(&R9) == t.device_of(&L9));
        CHECK(t.get(&R9)->device == t.device_of(&L9));
        // unknown addresses: no device, no move
        char stranger;
        CHECK(t.device_of(&stranger) == -1 && !t.colocate(&stranger, &L[0]) && !t.colocate(&R[0], &stranger));
        t.destroy_all();
        CHECK(g_live.empty());
    }
    // (g) eight mono cameras (Left + Ini each) declared from interleaving threads still end up spread over the devices; one device
    //     (the default: GFO_DEVICE or 0) never moves anything
    {
        gfo_adapter::ContextTable t;
        t.set_devices(std::vector<int>{0, 1, 2, 3, 4, 5, 6, 7});
        std::vector<char> objs(16);
        for (size_t i = 0; i < objs.size(); i++) { t.declare(&objs[i], prm); CHECK(t.get(&objs[i]) != NULL); }
        int per_dev[8] = {0};
        for (size_t i = 0; i < objs.size(); i++) per_dev[t.device_of(&objs[i])]++;
        for (int d = 0; d < 8; d++) CHECK(per_dev[d] == 2);
        // ... and were they eight STEREO cameras (objs[2k] left, objs[2k + 1] right), their first frames leave one rig per device
        for (int k = 0; k < 8; k++) { t.colocate(&objs[2 * k + 1], &objs[2 * k]); CHECK(t.device_of(&objs[2 * k + 1]) == t.device_of(&objs[2 * k])); }
        std::set<int> rig_dev;
        for (int k = 0; k < 8; k++) rig_dev.insert(t.device_of(&objs[2 * k]));
        CHECK(rig_dev.size() == 8 && t.moved() == 8);
        t.destroy_all();
        gfo_adapter::ContextTable one;
        unsetenv("GFO_DEVICES");
        setenv("GFO_DEVICE", "3", 1);
        char a, b;
        one.declare(&a, prm); one.declare(&b, prm);
        CHECK(one.device_count() == 1 && one.get(&a)->device == 3 && one.get(&b)->device == 3 && !one.colocate(&b, &a) && one.moved() == 0);
        one.destroy_all();
        setenv("GFO_DEVICES", "2, 2,5", 1);                 // the environment form, a device listed twice = two slots
        gfo_adapter::ContextTable env;
        char c0, c1, c2;
        env.declare(&c0, prm); env.declare(&c1, prm); env.declare(&c2, prm);
        CHECK(env.device_count() == 3 && env.device_of(&c0) == 2 && env.device_of(&c1) == 2 && env.device_of(&c2) == 5);
        CHECK(env.slot_of(&c0) == 0 && env.slot_of(&c1) == 1 && env.colocate(&c1, &c0) && env.slot_of(&c1) == env.slot_of(&c0));
        env.destroy_all();
        unsetenv("GFO_DEVICES"); unsetenv("GFO_DEVICE");
        CHECK(g_live.empty());
    }
    printf("OK created %d destroyed %d\n", g_created, g_destroyed);
    return 0;
}
