// lmh_device.h -- parameter block shared by the host launcher and the gfx950 kernels.
#pragma once
#include <stdint.h>

#define LMH_MODEL_STRIDE 400  // per model: 28 x 14 doubles (Ibar 9 | m*c 3 | m | Robot::desiredPosture of coordinate i, the record's spare slot) + [392] total mass
#define LMH_BODY_STRIDE 14
#define LMH_SEG_STRIDE 52
#define LMH_ROLLOUT_THREADS 128   // fused rollout: two waves per robot (lmh_kernels.hip, bsync)

struct LmhIkTarget { double v[16]; };   // rF(6) | lF(6) | com(3) | pad: passed by value in the IK kernel's arguments

struct LmhDevParams {
    // ---- device buffers
    const double *model;        // [n_models][LMH_MODEL_STRIDE]
    const double *mpc;          // [n_gain][mpc_stride]: K(N+1) | Px0(N+1) | Px1(N+1) | zcom | pad(3)
    const double *zmpx;         // [n_samples]
    const double *zmpy;
    const uint8_t *phase;       // [n_samples] or nullptr
    const double *gcol;         // [16][6] friction-cone generators of one foot | (G G')^-1 | G'(G G')^-1
    const double *segs;         // [n_seg][LMH_SEG_STRIDE] walking segments: t0 | rF[3][8] | lF[3][8] | pad, or nullptr
    const uint16_t *seg_of_sample; // [n_samples] segment of preview index k
    const double *xscale;       // [n_instances] per-instance scale of ZMP x and x-axis foot polynomials, or nullptr
    int32_t model_stride;       // 0 = shared model
    int32_t mpc_stride_inst;    // 0 = shared gain row
    int32_t mpc_stride;         // 3*(N+1)+4
    int32_t n_samples;
    int32_t n_seg;
    int32_t horizon;            // N
    int32_t n_instances;
    int32_t warm_start;
    int32_t max_qp_iters;
    int32_t precision;          // 0: fp64 throughout; 1: mixed (fp32 model terms, fp64 references + QP); 2: fp32
    int32_t bpp_max;            // block-pivoting rounds before the Lawson-Hanson pass (< 0: Lawson-Hanson only)
    int32_t plant;              // 1: compliant-contact plant driven by the torques (lmh_config.plant)
    // ---- scalars (reference literals, see include/lmh.h lmh_config)
    double dt;                  // control step (RK4 step of the fused rollout, Clock::step)
    double mpc_dt;              // MPC sample time: k = int(t / mpc_dt), sample period of the reference arrays (lmh_config.mpc_dt; = dt when that is 0)
    double kp_joints, kd_joints, kp_mom, kd_mom, kp_feet, kd_feet;
    double w_com_lin, w_com_ang, w_base_pos, w_base_ang, w_joints, w_force, w_foot;
    double eps_coeff;
    double inv_w_base_pos, inv_w_base_ang, inv_w_joints, inv_w_com_lin, inv_w_foot;   // 1.0 / weight, divided once on the host (the same IEEE quotient the kernels used to form per evaluation)
    double contact_k, contact_d, contact_dt, contact_mu;
    double a00, a01, a10, a11, b0, b1;   // LIPM A, B (mpcLinearPendulum.cpp:45-47)
    // ---- foot reference polynomials (shared), ascending powers
    double rF[3][8];
    double lF[3][8];
    int32_t rFn[3];
    int32_t lFn[3];
};

// walking-plan generator (lmh_gen_walk): arguments of the device kernel
#define LMH_GEN_MAX_STEPS 1022
struct LmhWalkSpec {
    double time_step, time_per_step, ds_time, step_height, settle_time, foot_y;
    int32_t n_samples, num_steps, first_support, pad;
};
