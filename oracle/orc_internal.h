/* orc_internal.h -- shared private declarations of the CPU oracle (test infrastructure). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "lmh_oracle.h"

void orc_cross_matrix(const double v[3], double A[9]);
void orc_velocity_matrix(const double T[16], double X[36]);
void orc_inverse_transform(const double T[16], double Ti[16]);
void orc_spatial_cross(const double v[6], double m[36]);
void orc_spatial_cross_force(const double v[6], double f[36]);
void orc_omega_to_euler_rate(const double eta[3], double Om[9]);
void orc_euler_to_so3(const double rpy[3], double R[9]);
void orc_rot_to_axis_angle(const double R[9], double r[3]);
double orc_polyval(const double *poly, int n, double x);
int orc_polyder(const double *poly, int n, double *out);
void orc_swap_base_velocity(const double X01[36], double v[ORC_NQ]);

#endif
