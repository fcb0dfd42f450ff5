// The stop / renew / confirm rules of the LOBPCG loop of scs_fiedler (scs_eig.hip), as one small host-side
// state machine -- factored out of the solver's host function in round 6 so that the thresholds can be
// unit-tested with scripted residual sequences (tests/test_boundary_cpu.py, through scs_debug_loop_policy;
// no device involved).  What replaces the reference's ARPACK stopping rule (eigsh with tol = 0: "to machine
// precision", scipy/sparse/linalg/_eigen/arpack/arpack.py:1359-1720 behind sklearn's
// _spectral_embedding.py:370-372) is a residual target with guards; the numbers below are those rules.
#pragma once

#include <algorithm>

struct scs_loop_policy {
    // ---- the rules' constants (scs_fiedler reads SCS_LOWP* into the three lowp_* fields)
    double tol = 1e-13;        // ||S x - lambda x|| target of the wanted pair(s)
    int lowp_mode = 2;         // 0 no image, 1 image up to the first renewal, 2 image throughout
    double lowp_tol = 1e-8;    // residual at which S X / S P are renewed through W (first renewal)
    double lowp_tol2 = 0.0;    // ... and a second time (0: only when the residual stops halving)
    static constexpr double HALVING = 0.5;        // "progress" = the residual fell below half the best seen
    static constexpr int STALL = 12;              // iterations without progress ...
    static constexpr double FLOOR = 1e-9;         // ... below this residual: the floating-point floor, stop
    static constexpr double AWAY = 100.0;         // residual this many times the best seen (< 1e-6): diverging
    static constexpr double AWAY_FROM = 1e-6;
    static constexpr int IMAGE_STALL = 8;         // after a renewal: the image's rounding has become the floor
    static constexpr int RENEW_STALL = 8;         // before it: a plateau below 1e-6 also asks for the renewal
    static constexpr double RENEW_STALL_BELOW = 1e-6;
    static constexpr int RENEW2_STALL = 4;        // second renewal: four iterations without progress below lowp_tol
    static constexpr int MAX_CONFIRMATIONS = 3;

    // ---- state
    double best_res = 1e300;
    int since_best = 0;
    int lowp_state = 0;  // image in use: 1 + the renewals made so far; 0: all double
    int confirmations = 0;

    enum { GO_ON = 0, STOP = 1, RENEW = 2 };

    // One iteration's residual (the worst of the wanted columns): what the loop does next.  STOP sends the loop
    // to the confirmation (X and S X renewed through W, the residual measured again: `confirm`); RENEW has S X and
    // S P computed anew through W behind the iteration already enqueued and the image stays (mode 2) or goes.
    int step(double worst) {
        bool stop = worst <= tol;
        if (worst < HALVING * best_res) {
            best_res = worst;
            since_best = 0;
        } else if (++since_best >= STALL && worst < FLOOR) {
            stop = true;  // stagnated at the floating-point floor
        } else if (best_res < AWAY_FROM && worst > AWAY * best_res) {
            stop = true;  // moving away from where it had been (a guard, not a path)
        }
        if (!stop && lowp_state >= 2 && since_best >= IMAGE_STALL) stop = true;
        int action = GO_ON;
        if (!stop && ((lowp_state == 1 && (worst <= lowp_tol || (since_best >= RENEW_STALL && worst < RENEW_STALL_BELOW))) ||
                      (lowp_state == 2 && (worst <= lowp_tol2 || (since_best >= RENEW2_STALL && worst < lowp_tol))))) {
            lowp_state = lowp_mode >= 2 ? lowp_state + 1 : 0;
            action = RENEW;
        }
        if (stop) {
            lowp_state = 0;  // the confirmation, and whatever follows it, through W
            action = STOP;
        }
        return action;
    }
    bool image_in_use() const { return lowp_state > 0; }
    // May another confirmation be made (at most MAX_CONFIRMATIONS per solve)?  Counts it.
    bool begin_confirmation() {
        if (confirmations >= MAX_CONFIRMATIONS) return false;
        ++confirmations;
        return true;
    }
    // The residual measured through W after a confirmation: true = the loop ends (`converged` says how), false =
    // it goes on in double precision with the search directions restarted.
    bool confirm(double w2, bool *converged) {
        if (w2 <= tol || (since_best >= STALL && w2 < FLOOR) || confirmations >= MAX_CONFIRMATIONS) {
            *converged = w2 <= tol;
            return true;
        }
        best_res = w2;
        since_best = 0;
        return false;
    }
};
