/* offline_stand.c -- CPU ORACLE (test infrastructure): the apps/offline workload
 * (reference apps/offline/main.cpp:12-89): IK to CoM (-0.02,0,0.26), stand for T seconds
 * under RK4, print the CoM x the reference prints after every tick (k4-stage Robot state).
 * usage: offline_stand [T=5] [dt=0.01] [horizon_s=0.5] [wbc_calls=1] [quiet=0] */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <time.h>
#include "lmh_oracle.h"

int main(int argc, char **argv)
{
    double T = argc > 1 ? atof(argv[1]) : 5.0;
    double dt = argc > 2 ? atof(argv[2]) : 0.01;
    double th = argc > 3 ? atof(argv[3]) : 0.5;
    int calls = argc > 4 ? atoi(argv[4]) : 1;
    int quiet = argc > 5 ? atoi(argv[5]) : 0;
    orc_system *s = (orc_system *)malloc(sizeof(orc_system));
    orc_eval *ev = (orc_eval *)malloc(sizeof(orc_eval));
    orc_system_init_offline(s, T, dt, th, 1);
    s->ctl.wbc_calls_per_eval = calls;
    s->mpc.faithful_rebuild = (calls > 1);
    double state[60];
    for (int i = 0; i < 30; i++) { state[i] = s->robot.q[i]; state[30 + i] = s->robot.v[i]; }
    double t = 0.0;
    int ticks = 0;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    while (fabs(t - T) > 0.01) {                /* main.cpp:74 */
        orc_rk4_tick(s, state, t, dt, ev);
        if (!quiet) printf("%.15g\n", s->robot.CoM[0]);
        t += dt;                                /* Clock::step */
        ticks++;
    }
    clock_gettime(CLOCK_MONOTONIC, &b);
    double sec = (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
    fprintf(stderr, "ticks=%d  wall=%.3fs  ticks/s=%.1f  final CoM x=%.9g  sum fz=%.9g  qp_iters(last)=%d\n",
            ticks, sec, ticks / sec, s->robot.CoM[0], ev->f[5] + ev->f[11], ev->qp_iters);
    orc_system_free(s);
    free(s); free(ev);
    return 0;
}
