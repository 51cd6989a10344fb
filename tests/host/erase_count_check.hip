// Host-only check of the pipelined driver's EARLY erase count (backend.hip: lvk_vio_pipe_submit) against the sequential rule it
// stands in for (batch_imu, larvio.cpp:464-517: the count taken when the update starts, with the td of that moment).
// Model: IMU stamps on a jittered grid with a random phase against the image grid, td a random walk, up to three updates between
// the td a count is taken from and the td the update starts with.  Checked: (1) whenever the count is the same for td_pub - margin
// and td_pub + margin, it is THE count for every td within the margin (and so is the state time it leaves behind); (2) the share
// of message frames that have to wait (a sample inside the margin) is what the geometry says, 2*margin / IMU period; (3) chained
// over a run - state time carried from count to count as the driver does - the early counts reproduce the sequential erase
// sequence exactly.  Prints "ok ..." or "MISMATCH ...".
#include "../../larvio_amd/csrc/backend.hip"
#include <random>

int main()
{
    std::mt19937_64 rng(2026);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    const double th = 1.0 / (2 * 200.0), margin = 5e-4, period = 0.005;
    long early = 0, waited = 0, msgs = 0;
    for (int run = 0; run < 200; ++run) {
        // IMU stream: 200 Hz with a random phase and +-50 us jitter; images every 50 ms, a message every other image
        const double phase = U(rng) * period;
        std::vector<lvk_imu> imu;
        for (int k = 0; k < 4000; ++k) { lvk_imu s; memset(&s, 0, sizeof s); s.t = phase + k * period + (U(rng) - 0.5) * 1e-4; imu.push_back(s); }
        double td = (U(rng) - 0.5) * 4e-3;                 // true time offset of this run
        double td_hist[4] = {td, td, td, td};             // td after the last finished updates (the driver publishes the newest it has)
        double t_state = imu[0].t - 1.0;                  // sequential filter's state time
        double t_mirror = t_state;                        // the driver's mirror of it
        size_t head_seq = 0, head_pipe = 0;
        for (int m = 0; m < 180; ++m) {
            const double ts = 0.3 + 0.1 * m;
            // the driver's IMU view of this message: everything pushed so far (t < ts + 0.05) from its head
            size_t end = head_seq; while (end < imu.size() && imu[end].t < ts + 0.05) ++end;
            // sequential rule: count with the td at the start of the update
            double ta_seq = 0;
            const int n_seq = imu_erase_count(t_state, ts + td, th, imu.data() + head_seq, (int)(end - head_seq), &ta_seq);
            // early rule: td published 0..3 updates ago
            const double td_pub = td_hist[rng() % 4];
            if (fabs(td - td_pub) > margin) { printf("MISMATCH model: td moved by more than the margin\n"); return 1; }
            double ta = 0, tb = 0;
            const int n_lo = imu_erase_count(t_mirror, ts + td_pub - margin, th, imu.data() + head_pipe, (int)(end - head_pipe), &ta);
            const int n_hi = imu_erase_count(t_mirror, ts + td_pub + margin, th, imu.data() + head_pipe, (int)(end - head_pipe), &tb);
            ++msgs;
            if (n_lo == n_hi) {
                ++early;
                if (n_lo != n_seq || ta != ta_seq || tb != ta_seq) { printf("MISMATCH run %d msg %d: early %d (t %.6f) sequential %d (t %.6f)\n", run, m, n_lo, ta, n_seq, ta_seq); return 1; }
            } else {
                ++waited;
                if (n_seq < n_lo || n_seq > n_hi) { printf("MISMATCH run %d msg %d: sequential count %d outside [%d, %d]\n", run, m, n_seq, n_lo, n_hi); return 1; }
            }
            // both schedules consume the sequential count (a waiting frame gets it from the filter's thread)
            head_seq += (size_t)n_seq; head_pipe += (size_t)n_seq; t_state = ta_seq; t_mirror = ta_seq;
            if (head_seq != head_pipe) { printf("MISMATCH heads\n"); return 1; }
            // the update moves td a little (a few microseconds; 40 us now and then)
            td += (U(rng) - 0.5) * ((rng() % 16) ? 6e-6 : 8e-5);
            td_hist[3] = td_hist[2]; td_hist[2] = td_hist[1]; td_hist[1] = td_hist[0]; td_hist[0] = td;
            const double lo = *std::min_element(td_hist, td_hist + 4), hi = *std::max_element(td_hist, td_hist + 4);
            if (hi - lo > margin) { td_hist[1] = td_hist[2] = td_hist[3] = td; }   // keep the model inside its own premise
        }
    }
    const double share = (double)waited / (double)msgs, expect = 2 * margin / period;
    if (fabs(share - expect) > 0.04) { printf("MISMATCH share of waiting frames %.3f, expected about %.3f\n", share, expect); return 1; }
    // the benchmark's input: stamps exactly on the image grid, td near zero -> the bound sits half a period from every sample: never waits
    {
        std::vector<lvk_imu> imu; for (int k = 0; k < 4000; ++k) { lvk_imu s; memset(&s, 0, sizeof s); s.t = k * period; imu.push_back(s); }
        double t_state = -1; size_t head = 0;
        for (int m = 0; m < 150; ++m) {
            const double ts = 0.3 + 0.1 * m, td = 1e-5 * ((m % 7) - 3);
            size_t end = head; while (end < imu.size() && imu[end].t < ts + 0.05) ++end;
            double ta = 0, tb = 0;
            const int n_lo = imu_erase_count(t_state, ts + td - margin, th, imu.data() + head, (int)(end - head), &ta);
            const int n_hi = imu_erase_count(t_state, ts + td + margin, th, imu.data() + head, (int)(end - head), &tb);
            if (n_lo != n_hi) { printf("MISMATCH grid-aligned input would wait at message %d\n", m); return 1; }
            head += (size_t)n_lo; t_state = ta;
        }
    }
    printf("ok %ld messages, %ld counted early, %ld waiting (%.1f %%)\n", msgs, early, waited, 100.0 * share);
    return 0;
}
