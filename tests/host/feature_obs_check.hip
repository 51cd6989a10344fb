// Host-only check of backend.hip's per-feature observation list (std::map<StateIDType, ...> in the reference, a sorted vector with
// newest-first / bisection lookups here) and of add_observations' cursor walk over the feature map: random operation sequences against
// plain std::map models.  Prints "ok" or "MISMATCH ...".
#include "../../larvio_amd/csrc/backend.hip"
#include <random>

static int check_feature_obs()
{
    std::mt19937_64 rng(7);
    for (int rep = 0; rep < 200; ++rep) {
        Feature f; std::map<long long, double> model;
        long long next = 100;
        for (int step = 0; step < 400; ++step) {
            const int op = (int)(rng() % 10);
            if (op < 5) { next += 1 + (long long)(rng() % 2); f.set(next, (double)next, 0, 0, 0); model[next] = (double)next; }          // newest (the normal case)
            else if (op < 6) { const long long sid = 100 + (long long)(rng() % (next - 99 + 3)); f.set(sid, (double)sid + 0.5, 0, 0, 0); model[sid] = (double)sid + 0.5; }   // out of order / overwrite
            else if (op < 8 && !model.empty()) { auto it = model.begin(); std::advance(it, (long)(rng() % model.size())); f.erase(it->first); model.erase(it); }
            else { const long long sid = 99 + (long long)(rng() % (next - 98 + 4)); const int i = f.find(sid); auto it = model.find(sid);
                   if ((i >= 0) != (it != model.end()) || (i >= 0 && f.obs[i].z[0] != it->second)) { printf("MISMATCH find sid %lld\n", sid); return 1; } }
            if (f.obs.size() != model.size()) { printf("MISMATCH size\n"); return 1; }
            size_t k = 0; for (auto& kv : model) { if (f.obs[k].sid != kv.first || f.obs[k].z[0] != kv.second) { printf("MISMATCH order at %zu\n", k); return 1; } ++k; }
        }
    }
    return 0;
}

static int check_add_observations()
{
    std::mt19937_64 rng(11);
    lvk_ekf* e = new lvk_ekf(); e->ctx = nullptr; memset(&e->cfg, 0, sizeof e->cfg);
    std::map<long long, int> model;                       // id -> number of observations
    long long next_id = 1;
    std::vector<long long> live;
    for (int msg = 0; msg < 300; ++msg) {
        e->imu_id = 1000 + msg; e->imu_dt = 0.0;
        Clone c; memset(&c, 0, sizeof c); c.id = e->imu_id; e->clones.push_back(c); if (e->clones.size() > 12) e->clones.erase(e->clones.begin()); e->ranks_dirty = true;
        // tracks die at random, new ones are appended with growing ids; now and then the message is shuffled (ids out of order)
        std::vector<long long> keep; for (long long id : live) if (rng() % 8) keep.push_back(id);
        const int n_new = (int)(rng() % 6); for (int k = 0; k < n_new; ++k) keep.push_back(next_id++);
        live = keep;
        std::vector<long long> order = live; if (msg % 17 == 5) std::shuffle(order.begin(), order.end(), rng);
        std::vector<lvk_feature_obs> feats(order.size());
        for (size_t i = 0; i < order.size(); ++i) { memset(&feats[i], 0, sizeof feats[i]); feats[i].id = (uint64_t)order[i]; feats[i].u = (double)order[i]; feats[i].v = msg; feats[i].u_init = -1; feats[i].v_init = -1; }
        add_observations(e, feats.data(), (int)feats.size());
        for (long long id : order) model[id] += 1;
        if (msg % 5 == 4 && !e->map.empty()) {            // erase a few features, as the filter does after using them
            for (int k = 0; k < 3 && !e->map.empty(); ++k) { auto it = e->map.begin(); std::advance(it, (long)(rng() % e->map.size())); const long long id = it->first; e->map.erase(it); model.erase(id);
                                                              live.erase(std::remove(live.begin(), live.end(), id), live.end()); }
        }
        if (e->map.size() != model.size()) { printf("MISMATCH map size %zu vs %zu at message %d\n", e->map.size(), model.size(), msg); return 1; }
        auto a = e->map.begin(); auto b = model.begin();
        for (; a != e->map.end(); ++a, ++b) {
            if (a->first != b->first || a->second.id != b->first || (int)a->second.obs.size() != b->second || a->second.total_obs != b->second) { printf("MISMATCH feature %lld at message %d\n", a->first, msg); return 1; }
            if (a->second.obs.back().sid != e->imu_id && std::find(order.begin(), order.end(), a->first) != order.end()) { printf("MISMATCH newest sid\n"); return 1; }
        }
    }
    return 0;
}

// FeatureMap (the id-ordered flat container that replaced std::map<id, Feature>) against a std::map model: random emplace / operator[] /
// erase by id and by iterator / find / at / purge, ids mostly growing with a few out of order, erased ids re-created before and after a
// purge; after every step size, order, liveness and object addresses of untouched features must agree.
static int check_feature_map()
{
    std::mt19937_64 rng(23);
    for (int rep = 0; rep < 60; ++rep) {
        FeatureMap fm; std::map<long long, std::pair<int, const Feature*>> model;      // id -> (total_obs tag, address)
        long long next = 10;
        for (int step = 0; step < 1500; ++step) {
            const int op = (int)(rng() % 16);
            if (op < 6) {                                                             // new track: growing id (append)
                const long long id = next; next += 1 + (long long)(rng() % 3);
                Feature& f = fm[id]; f.total_obs = (int)id;
                if (f.id != id || !f.obs.empty() || f.in_state) { printf("MISMATCH fresh feature %lld\n", id); return 1; }
                model[id] = {(int)id, &f};
            } else if (op < 8) {                                                      // out-of-order / existing / erased id through emplace
                const long long id = 10 + (long long)(rng() % (unsigned long long)(next - 9));
                const bool had = model.count(id) != 0;
                auto pr = fm.emplace(id, Feature());
                if (pr.second == had || pr.first->first != id) { printf("MISMATCH emplace %lld\n", id); return 1; }
                if (!had) { pr.first->second.total_obs = (int)id; if (!pr.first->second.obs.empty()) { printf("MISMATCH revived feature not reset\n"); return 1; } model[id] = {(int)id, &pr.first->second}; }
            } else if (op < 12 && !model.empty()) {                                    // erase by id
                auto it = model.begin(); std::advance(it, (long)(rng() % model.size()));
                if (fm.erase(it->first) != 1 || fm.erase(it->first) != 0) { printf("MISMATCH erase %lld\n", it->first); return 1; }
                model.erase(it);
            } else if (op < 13 && !model.empty()) {                                    // erase by iterator
                auto it = fm.begin(); std::advance(it, (long)(rng() % fm.size()));
                const long long id = it->first; fm.erase(it); model.erase(id);
            } else if (op < 14) {
                fm.purge();
            } else {                                                                  // lookups, present or not
                const long long id = 9 + (long long)(rng() % (unsigned long long)(next - 7));
                const bool had = model.count(id) != 0;
                if ((fm.find(id) != fm.end()) != had) { printf("MISMATCH find %lld\n", id); return 1; }
                bool threw = false; try { (void)fm.at(id); } catch (const std::out_of_range&) { threw = true; }
                if (threw == had) { printf("MISMATCH at %lld\n", id); return 1; }
            }
            if (fm.size() != model.size() || fm.empty() != model.empty()) { printf("MISMATCH size %zu vs %zu\n", fm.size(), model.size()); return 1; }
            auto b = model.begin();
            for (auto kv : fm) {
                if (b == model.end() || kv.first != b->first || kv.second.id != b->first || kv.second.total_obs != b->second.first || &kv.second != b->second.second) { printf("MISMATCH walk at id %lld\n", kv.first); return 1; }
                ++b;
            }
            if (b != model.end()) { printf("MISMATCH walk ended early\n"); return 1; }
        }
    }
    return 0;
}

int main() { int r = check_feature_obs(); r |= check_add_observations(); r |= check_feature_map(); if (!r) printf("ok\n"); return r; }
