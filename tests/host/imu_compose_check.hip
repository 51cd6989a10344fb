// Host-only check of backend.hip's IMU composition (no GPU needed; the HIP runtime is linked but never called):
// compose_transition<L> (structure-aware, vectorisable loop nests) must equal the plain dense recurrences
//     Phi_tot <- Phi Phi_tot ;  Q_tot <- Phi Q_tot Phi^T + (Phi G) Qc (Phi G)^T dt
// bit for bit (larvio.cpp:520-578 does them densely).  Prints "ok <checksum>" or "MISMATCH ...".
#include "../../larvio_amd/csrc/backend.hip"

template <int L> static void dense_step(const lvk_ekf* e, const double* Phi, double dt, const double* C, double* Pt, double* Qt, bool first)
{
    double G[L * 12]; memset(G, 0, sizeof G);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { G[i * 12 + j] = -C[i * 3 + j]; G[(3 + i) * 12 + 3 + j] = -C[i * 3 + j]; }
    for (int i = 0; i < 3; ++i) { G[(9 + i) * 12 + 6 + i] = 1.0; G[(12 + i) * 12 + 9 + i] = 1.0; }
    static double PG[L * 12], Q[L * L], T[L * L], U[L * L];
    for (int i = 0; i < L; ++i) for (int j = 0; j < 12; ++j) { double s = 0; for (int k = 0; k < L; ++k) s += Phi[i * L + k] * G[k * 12 + j]; PG[i * 12 + j] = s; }
    for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) { double s = 0; for (int k = 0; k < 12; ++k) s += PG[i * 12 + k] * e->Qc[k] * PG[j * 12 + k]; Q[i * L + j] = s * dt; }
    if (first) { memcpy(Pt, Phi, sizeof(double) * L * L); memcpy(Qt, Q, sizeof(double) * L * L); return; }
    for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) { double s = 0; for (int k = 0; k < L; ++k) s += Phi[i * L + k] * Pt[k * L + j]; T[i * L + j] = s; }
    memcpy(Pt, T, sizeof T);
    for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) { double s = 0; for (int k = 0; k < L; ++k) s += Phi[i * L + k] * Qt[k * L + j]; T[i * L + j] = s; }
    for (int i = 0; i < L; ++i) for (int j = 0; j < L; ++j) { double s = 0; for (int k = 0; k < L; ++k) s += T[i * L + k] * Phi[j * L + k]; U[i * L + j] = s; }
    for (int i = 0; i < L * L; ++i) Qt[i] = U[i] + Q[i];
}

template <int L, bool CALIB> static int run()
{
    lvk_ekf* e = new lvk_ekf();
    e->ctx = nullptr; memset(&e->cfg, 0, sizeof e->cfg); e->cfg.calib_imu_instrinsic = CALIB ? 1 : 0; e->leg = L;
    memset(&e->s, 0, sizeof e->s); e->s.q[3] = 1.0; e->s_fej_now = e->s; e->s_old = e->s;
    for (int i = 0; i < 12; ++i) e->Qc[i] = 1e-4 * (1 + i);
    for (int i = 0; i < 9; ++i) { e->Tg[i] = e->Ma[i] = (i % 4 == 0) ? 1.0 + 0.01 * i : 0.002 * i; e->As[i] = 0.001 * (i + 1); }
    e->imu_img_time_th = 0.0025;
    static double Pt[L * L], Qt[L * L];
    double worst = 0, chk = 0;
    for (int rep = 0; rep < 40; ++rep) {
        e->have_prop = false;
        bool first = true;
        for (int i = 0; i < 20; ++i) {
            double g[3], a[3];
            for (int k = 0; k < 3; ++k) { g[k] = 0.3 * sin(0.37 * i + k + rep); a[k] = (k == 2 ? 9.8 : 0.0) + 0.5 * cos(0.21 * i + 2 * k + rep); }
            memcpy(e->m_gyro_old, g, 24); memcpy(e->m_acc_old, a, 24);
            g[0] += 0.01; a[1] -= 0.02;
            // the same sequence of calls process_model makes, with the dense recurrence run beside compose_transition
            const double t = e->s.t + 0.005;
            double f[3], w[3], w_old[3], f_old[3], acc[3], gyro[3], acc_old[3], gyro_old[3];
            for (int k = 0; k < 3; ++k) { f[k] = a[k] - e->s.ba[k]; f_old[k] = e->m_acc_old[k] - e->s.ba[k]; }
            if (CALIB) {
                double tt[3];
                m3_v(e->Ma, f, acc); m3_v(e->As, acc, tt); for (int k = 0; k < 3; ++k) w[k] = g[k] - tt[k] - e->s.bg[k]; m3_v(e->Tg, w, gyro);
                m3_v(e->Ma, f_old, acc_old); m3_v(e->As, acc_old, tt); for (int k = 0; k < 3; ++k) w_old[k] = e->m_gyro_old[k] - tt[k] - e->s.bg[k]; m3_v(e->Tg, w_old, gyro_old);
            } else for (int k = 0; k < 3; ++k) { w[k] = g[k] - e->s.bg[k]; w_old[k] = e->m_gyro_old[k] - e->s.bg[k]; acc[k] = f[k]; gyro[k] = w[k]; acc_old[k] = f_old[k]; gyro_old[k] = w_old[k]; }
            const double dt = t - e->s.t;
            predict_new_state(e, dt, gyro, acc);
            static double Phi[LEG_MAX * LEG_MAX];
            if (CALIB) cal_phi_calib(e, Phi, dt, f, w, acc, gyro, f_old, w_old, acc_old, gyro_old); else cal_phi(e, Phi, dt, w, w_old);
            double C[9]; quat_to_rot(e->s_old.q, C);
            dense_step<L>(e, Phi, dt, C, Pt, Qt, first); first = false;
            compose_transition<L, CALIB>(e, Phi, dt);
            e->s.t = t; e->s_fej_now.t = t;
            for (int k = 0; k < L * L; ++k) {
                if (memcmp(&Pt[k], &e->Phi_tot[k], 8) != 0 && !(Pt[k] == 0.0 && e->Phi_tot[k] == 0.0)) { printf("MISMATCH Phi L=%d rep %d step %d idx %d: %.17g vs %.17g\n", L, rep, i, k, Pt[k], e->Phi_tot[k]); return 1; }
                if (memcmp(&Qt[k], &e->Q_tot[k], 8) != 0 && !(Qt[k] == 0.0 && e->Q_tot[k] == 0.0)) { printf("MISMATCH Q L=%d rep %d step %d idx %d: %.17g vs %.17g\n", L, rep, i, k, Qt[k], e->Q_tot[k]); return 1; }
                chk += Pt[k] + Qt[k];
            }
        }
    }
    (void)worst;
    printf("ok L=%d %.17g\n", L, chk);
    return 0;
}
int main() { int r = run<22, false>(); r |= run<46, true>(); return r; }
