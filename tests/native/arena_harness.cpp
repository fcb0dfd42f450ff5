// CPU harness of the device-memory arena's logic (csrc/scs_arena.h) for tests/test_arena_cpu.py: the backing
// allocator is malloc with a budget, the "streams" of an owner are counters this program advances by script.
//   arena_harness random SEED STEPS   -> random allocations / releases / polls with the invariants checked
//   arena_harness scenario            -> the fixed scenarios (carving, coalescing, pending chunks, trimming)
// Prints "ok ..." lines; exits non-zero with a message on the first violated expectation.
#include "../../spectralclustersupertree_amd/csrc/scs_arena.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <set>

namespace {
struct fake_owner {
    // work queued on the owner's one stream: `queued` counts submissions, `done` completions
    uint64_t queued = 0, done = 0;
};
struct fake_event {
    fake_owner *owner;
    uint64_t at;
};

struct world {
    scs_arena_core a;
    size_t budget, taken = 0;
    std::map<void *, size_t> driver;  // live driver allocations
    std::set<fake_event *> live_events;
    int recycled = 0;

    explicit world(size_t budget_bytes) : budget(budget_bytes) {
        a.back_alloc = [this](size_t bytes) -> void * {
            if (taken + bytes > budget) return nullptr;
            // (addresses only: the arena never touches the memory.  Reserve address space without committing it.)
            void *p = aligned_alloc(4096, 4096);
            static char *next = (char *)0x100000000000ull;
            free(p);
            p = next;
            next += (bytes + 0xFFFFF) / 0x100000 * 0x100000 + 0x100000;
            taken += bytes;
            driver[p] = bytes;
            return p;
        };
        a.back_free = [this](void *p) {
            auto it = driver.find(p);
            if (it == driver.end()) {
                printf("FAIL: the arena handed back %p, which the driver never gave\n", p);
                exit(1);
            }
            taken -= it->second;
            driver.erase(it);
        };
        a.owner_mark = [this](const void *owner) {
            fake_owner *o = (fake_owner *)owner;
            std::vector<void *> ev;
            if (o->done < o->queued) {
                auto *e = new fake_event{o, o->queued};
                live_events.insert(e);
                ev.push_back(e);
            }
            return ev;
        };
        a.event_done = [](void *e) {
            auto *f = (fake_event *)e;
            return f->owner->done >= f->at;
        };
        a.event_wait = [](void *e) {
            auto *f = (fake_event *)e;
            if (f->owner->done < f->at) f->owner->done = f->at;  // (waiting lets the stream run)
        };
        a.event_recycle = [this](void *e) {
            live_events.erase((fake_event *)e);
            delete (fake_event *)e;
            ++recycled;
        };
    }
};

#define EXPECT(cond, ...)                  \
    do {                                   \
        if (!(cond)) {                     \
            printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            printf(__VA_ARGS__);           \
            printf("\n");                  \
            exit(1);                       \
        }                                  \
    } while (0)

void scenarios() {
    const size_t MB = 1 << 20, GB = 1 << 30;
    {  // carving and coalescing inside one slab; the driver is asked once
        world w(64 * GB);
        fake_owner me;
        void *a = w.a.alloc(8 * GB, &me);
        EXPECT(a && w.a.n_driver_allocs == 1 && w.a.slab_bytes == 8 * GB, "one slab of the size asked for");
        EXPECT(w.a.release(a), "release");
        void *b = w.a.alloc(3 * GB, &me), *c = w.a.alloc(2 * GB, &me), *d = w.a.alloc(3 * GB - 4096, &me);
        EXPECT(b && c && d && w.a.n_driver_allocs == 1, "three chunks carved out of the released slab (%llu driver calls)",
               (unsigned long long)w.a.n_driver_allocs);
        EXPECT(b == a && (char *)c == (char *)b + 3 * GB, "best fit from the front");
        EXPECT(w.a.check(), "consistent");
        w.a.release(c);
        w.a.release(b);
        w.a.release(d);
        EXPECT(w.a.check() && w.a.used_bytes == 0, "all released");
        void *e = w.a.alloc(8 * GB, &me);
        EXPECT(e == a && w.a.n_driver_allocs == 1, "the pieces merged back into the whole slab");
        w.a.release(e);
        EXPECT(w.a.trim(0) == 8 * GB && w.taken == 0 && w.a.slab_bytes == 0, "trim hands the free slab back");
        printf("ok carve/coalesce/trim\n");
    }
    {  // small requests live in slabs of their own and never pin a large slab
        world w(64 * GB);
        fake_owner me;
        void *big = w.a.alloc(4 * GB, &me);
        void *s1 = w.a.alloc(100, &me), *s2 = w.a.alloc(70000, &me);
        EXPECT(w.a.n_driver_allocs == 2 && w.taken == 4 * GB + scs_arena_core::SMALL_SLAB, "one large slab, one small slab");
        EXPECT((char *)s2 == (char *)s1 + 512, "small grain 512 bytes");
        w.a.release(big);
        EXPECT(w.a.trim(0) == 4 * GB, "the large slab goes back although small chunks are alive");
        w.a.release(s1);
        w.a.release(s2);
        EXPECT(w.a.trim(0) == scs_arena_core::SMALL_SLAB && w.taken == 0, "then the small one");
        printf("ok small/large slabs\n");
    }
    {  // a released chunk is its owner's at once, another context's only when the owner's stream has passed
        world w(64 * GB);
        fake_owner a_ctx, b_ctx;
        void *p = w.a.alloc(2 * GB, &a_ctx);
        a_ctx.queued = 5;  // kernels on p in flight
        w.a.release(p);
        EXPECT(w.a.n_pending == 1, "pending");
        void *q = w.a.alloc(2 * GB, &b_ctx);
        EXPECT(q && q != p && w.a.n_driver_allocs == 2, "the other context does not get the pending chunk");
        void *r = w.a.alloc(1 * GB, &a_ctx);
        EXPECT(r == p && w.a.n_pending == 1 && w.a.n_pending_reuse == 1, "its owner does, carved; the rest stays pending");
        w.a.release(q);
        EXPECT(w.a.check(), "consistent");
        // the owner's stream completes: the next poll makes the rest everybody's
        a_ctx.done = 5;
        b_ctx.done = b_ctx.queued;
        void *t = w.a.alloc(3 * GB - 4096, &b_ctx);  // needs q's 2 GB? no: larger than any single free chunk
        EXPECT(t && w.a.n_driver_allocs == 3, "no free chunk of 3 GB: a third slab");
        void *u = w.a.alloc(1 * GB, &b_ctx);
        EXPECT(u == (char *)p + 1 * GB, "after the poll the rest of the first slab serves the other context");
        EXPECT(w.live_events.empty() || w.a.check(), "consistent");
        w.a.release(t);
        w.a.release(u);
        w.a.release(r);
        a_ctx.done = a_ctx.queued;
        EXPECT(w.a.trim(0) == 7 * GB - 4096 && w.taken == 0, "everything goes back (%zu left)", w.taken);
        EXPECT(w.live_events.empty(), "no event left behind");
        printf("ok pending chunks\n");
    }
    {  // a busy owner: the marker recorded by a poll decides, not the state at the release
        world w(64 * GB);
        fake_owner a_ctx, b_ctx;
        void *p = w.a.alloc(1 * GB, &a_ctx);
        a_ctx.queued = 3;
        w.a.release(p);
        a_ctx.queued = 9;  // more work queued behind (it does not use p)
        void *q = w.a.alloc(1 * GB, &b_ctx);
        EXPECT(q != p && w.live_events.size() == 1, "a marker at 9 is out");
        a_ctx.done = 3;  // the work on p is over, the marker's is not: conservative
        void *r = w.a.alloc(1 * GB, &b_ctx);
        EXPECT(r != p, "still not safe by what the arena can know");
        a_ctx.done = 9;
        void *s = w.a.alloc(1 * GB, &b_ctx);
        EXPECT(s == p && w.live_events.empty(), "the marker has completed");
        printf("ok markers\n");
    }
    {  // the driver refuses: everything released is waited for, free slabs go back, the request is tried again
        world w(10 * GB);
        fake_owner a_ctx, b_ctx;
        void *p = w.a.alloc(6 * GB, &a_ctx);
        a_ctx.queued = 2;
        w.a.release(p);
        void *q = w.a.alloc(3 * GB, &b_ctx);
        EXPECT(q != p, "a second slab (3 GB) while 6 GB are pending");
        void *r = w.a.alloc(5 * GB, &b_ctx);
        EXPECT(r == p && a_ctx.done == 2, "the refusal made the arena wait for the owner and reuse the 6 GB slab");
        void *t = w.a.alloc(4 * GB, &b_ctx);
        EXPECT(t == nullptr, "no room at all: nullptr");
        w.a.release(r);
        w.a.release(q);
        b_ctx.done = b_ctx.queued;
        void *u = w.a.alloc(8 * GB, &a_ctx);
        EXPECT(u && w.a.n_driver_frees >= 2, "free slabs went back to make room for one of 8 GB");
        printf("ok driver refusal\n");
    }
    {  // a context goes away
        world w(64 * GB);
        fake_owner a_ctx, b_ctx;
        void *p = w.a.alloc(1 * GB, &a_ctx), *q = w.a.alloc(1 * GB, &a_ctx), *keep = w.a.alloc(1 * GB, &b_ctx);
        a_ctx.queued = 1;
        w.a.release(p);
        w.a.owner_gone(&a_ctx);
        EXPECT(w.a.check() && w.a.used_bytes == 2 * GB && w.a.n_pending == 0,
               "what it had released is everybody's; what an object made on it still holds stays (orphaned)");
        EXPECT(w.a.holds(keep) && w.a.holds(q), "held chunks survive their context");
        void *r = w.a.alloc(1 * GB, &b_ctx);
        EXPECT(r == p, "the released one serves the other context at once");
        EXPECT(w.a.release(q) && w.a.n_pending == 0, "an orphan's release needs no stream: ready at once");
        EXPECT(!w.a.release(q), "and happens once");
        printf("ok owner_gone\n");
    }
    (void)MB;
}

void random_run(unsigned seed, int steps) {
    std::mt19937_64 rng(seed);
    world w((size_t)48 << 30);
    fake_owner owners[3];
    struct live { void *p; size_t bytes; int owner; };
    std::vector<live> held;
    std::map<char *, size_t> spans;  // in use: no overlap allowed
    // released memory: who released it and how much work its stream had queued then -- nobody else may get it
    // before that work is done
    struct rel { size_t bytes; int owner; uint64_t at; };
    std::map<char *, rel> released;
    int refused = 0, handed_over = 0;
    for (int step = 0; step < steps; ++step) {
        const int what = (int)(rng() % 100);
        if (what < 50 || held.empty()) {
            size_t bytes;
            switch (rng() % 4) {
            case 0: bytes = 1 + rng() % 4096; break;
            case 1: bytes = 4096 + rng() % (1 << 20); break;
            case 2: bytes = (1 << 20) + rng() % (64 << 20); break;
            default: bytes = ((size_t)64 << 20) + rng() % ((size_t)4 << 30); break;
            }
            const int o = (int)(rng() % 3);
            void *p = w.a.alloc(bytes, &owners[o]);
            if (!p) {
                ++refused;
                continue;
            }
            const size_t got = scs_arena_core::round_up(bytes);
            auto next = spans.lower_bound((char *)p);
            EXPECT(next == spans.end() || (char *)p + got <= next->first, "step %d: overlaps the chunk after it", step);
            if (next != spans.begin()) {
                auto prev = std::prev(next);
                EXPECT(prev->first + prev->second <= (char *)p, "step %d: overlaps the chunk before it", step);
            }
            {
                char *lo = (char *)p, *hi = lo + got;
                auto it = released.lower_bound(lo);
                if (it != released.begin() && std::prev(it)->first + std::prev(it)->second.bytes > lo) --it;
                while (it != released.end() && it->first < hi) {
                    const rel r = it->second;
                    char *rlo = it->first, *rhi = rlo + r.bytes;
                    if (r.owner != o) {
                        ++handed_over;
                        EXPECT(owners[r.owner].done >= r.at, "step %d: memory released by context %d (work up to %llu queued, %llu done) "
                               "handed to context %d", step, r.owner, (unsigned long long)r.at,
                               (unsigned long long)owners[r.owner].done, o);
                    }
                    it = released.erase(it);
                    if (rlo < lo) released[rlo] = {(size_t)(lo - rlo), r.owner, r.at};
                    if (rhi > hi) it = released.insert({hi, {(size_t)(rhi - hi), r.owner, r.at}}).first, ++it;
                }
            }
            spans[(char *)p] = got;
            held.push_back({p, bytes, o});
            owners[o].queued += rng() % 3;  // work on it
        } else if (what < 90) {
            const size_t i = rng() % held.size();
            EXPECT(w.a.release(held[i].p), "step %d: release", step);
            EXPECT(!w.a.release(held[i].p), "step %d: a second release must be refused", step);
            released[(char *)held[i].p] = {spans[(char *)held[i].p], held[i].owner, owners[held[i].owner].queued};
            spans.erase((char *)held[i].p);
            held[i] = held.back();
            held.pop_back();
        } else if (what < 96) {
            fake_owner &o = owners[rng() % 3];
            o.done = o.done + (o.queued - o.done) * (rng() % 3) / 2;  // the stream advances (or drains)
        } else if (what < 98) {
            if (w.a.trim((size_t)(rng() % 8) << 30)) {
                // (memory that went back to the driver: whoever gets those addresses again gets them from the driver)
                for (auto it = released.begin(); it != released.end();) {
                    bool live = false;
                    for (auto &sl : w.a.slabs)
                        if (sl.base && it->first >= sl.base && it->first < sl.base + sl.bytes) live = true;
                    it = live ? std::next(it) : released.erase(it);
                }
            }
        } else {
            w.a.poll(false);
        }
        if (step % 64 == 0) EXPECT(w.a.check(), "step %d: inconsistent maps", step);
        size_t used = 0;
        if (step % 256 == 0) {
            for (auto &h : held) used += scs_arena_core::round_up(h.bytes);
            EXPECT(used <= w.a.used_bytes, "step %d: used bytes", step);
        }
    }
    for (auto &h : held) w.a.release(h.p);
    for (auto &o : owners) o.done = o.queued;
    w.a.trim(0);
    EXPECT(w.a.check() && w.taken == 0 && w.a.slab_bytes == 0 && w.live_events.empty(),
           "at the end everything is back with the driver (%zu bytes, %zu events left)", w.taken, w.live_events.size());
    printf("ok random seed %u: %d steps, %llu requests, %llu driver allocations, %llu driver releases, %llu reuses of pending "
           "chunks, %llu polls, %d refused, %d times memory changed hands\n",
           seed, steps, (unsigned long long)w.a.n_allocs, (unsigned long long)w.a.n_driver_allocs,
           (unsigned long long)w.a.n_driver_frees, (unsigned long long)w.a.n_pending_reuse, (unsigned long long)w.a.n_polls, refused, handed_over);
}
}  // namespace

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "scenario")) {
        scenarios();
        return 0;
    }
    if (argc >= 4 && !strcmp(argv[1], "random")) {
        random_run((unsigned)atoi(argv[2]), atoi(argv[3]));
        return 0;
    }
    printf("usage: arena_harness scenario | random SEED STEPS\n");
    return 2;
}
