/*
 * target_manager_c.h -- the drop-in C boundary of the MI355X-native Kalman path.
 *
 * These are the ten entry points of the reference's C wrapper, with identical names, argument
 * order and types, so that a caller built against the reference's libtarget_c links against
 * libtarget_estimation_amd.so unchanged.  Each declaration cites the reference interface it
 * replaces: declaration in include/target_estimation/target_manager_c.h, definition in
 * src/target_manager_c.cpp (paths relative to the reference tree).
 *
 * Conventions (reference: include/target_estimation/target_manager.hpp:60):
 *   pose / measurement : double[7] = [x y z qx qy qz qw]
 *   twist              : double[6] = [vx vy vz wx wy wz]
 *   acceleration       : double[6] = [ax ay az alphax alphay alphaz]
 * All arrays are caller-owned; the library never keeps the pointers.
 *
 * Behavioural notes versus the reference (details in INTEGRATION.md):
 *   - every filter step executes on the GPU.  The one-target update calls are queued and executed
 *     as one indexed launch at the next call that reads or touches the batch (a getter, a batched
 *     call, target_manager_synchronize); per-target order is preserved, results are identical.  The
 *     batched entry points of target_batch_c.h are the fast path;
 *   - target_manager_new returns NULL (after printing the reason) where the reference lets a
 *     C++ exception escape through extern "C" (src/target_manager.cpp:114-115);
 *   - get_est_* leave the output array untouched for an unknown id (the reference copies a
 *     stale file-static temporary, src/target_manager_c.cpp:40-42) and are thread-safe.
 */
#ifndef TARGET_ESTIMATION_AMD_TARGET_MANAGER_C_H
#define TARGET_ESTIMATION_AMD_TARGET_MANAGER_C_H

#ifndef __cplusplus
#include <stdbool.h>
#endif

typedef void target_manager_c; /* opaque handle; reference: target_manager_c.h:20 */

#ifdef __cplusplus
extern "C" {
#endif

/* Create a manager whose default model (type, Q, R, P0) comes from a YAML model file.
 * Replaces target_manager_c.h:28 / target_manager_c.cpp:15-18 (TargetManager(file),
 * target_manager.cpp:111-118).  Precision f64; see target_manager_new_ex for f32. */
target_manager_c * target_manager_new(const char* file);

/* Create target `id` at pose p0 with zero twist/acceleration, filter time t0.  A second init of
 * an existing id prints "Target(id) already exists!" and changes nothing.
 * Replaces target_manager_c.h:29 / target_manager_c.cpp:20-24 (note the argument order
 * id, dt0, p0, t0) -> TargetManager::init(id,dt0,t0,p0), target_manager.cpp:135-179. */
void target_manager_init(const target_manager_c *self, const unsigned int id, const double dt0, double p0[], const double t0);

/* Predict by dt and correct with the measured pose.  Unknown id: prints
 * "Target(id) does not exist!" and returns.
 * Replaces target_manager_c.h:30 / target_manager_c.cpp:26-30 -> TargetManager::update(id,dt,meas),
 * target_manager.cpp:190-202 -> <Model>::addMeasurement (src/types). */
void target_manager_update_meas(const target_manager_c *self, const unsigned int id, const double dt, double meas[]);

/* Predict only.  Replaces target_manager_c.h:31 / target_manager_c.cpp:32-35 ->
 * TargetManager::update(id,dt), target_manager.cpp:204-218 -> <Model>::update(dt). */
void target_manager_update(const target_manager_c *self, const unsigned int id, const double dt);

/* Estimated pose.  Returns false for an unknown id.
 * Replaces target_manager_c.h:32 / target_manager_c.cpp:37-43 -> getTargetPose,
 * target_manager.cpp:252-261 -> TargetInterface::getEstimatedPose(), target_interface.cpp:100-104. */
bool target_manager_get_est_pose(const target_manager_c *self, const unsigned int id, double pose[]);

/* Estimated twist.  Replaces target_manager_c.h:33 / target_manager_c.cpp:45-51 ->
 * getTargetTwist, target_manager.cpp:263-272. */
bool target_manager_get_est_twist(const target_manager_c *self, const unsigned int id, double twist[]);

/* Estimated acceleration.  Replaces target_manager_c.h:34 / target_manager_c.cpp:53-59 ->
 * getTargetAcceleration, target_manager.cpp:274-283. */
bool target_manager_get_est_acceleration(const target_manager_c *self, const unsigned int id, double acceleration[]);

/* Number of measurements fused so far (0 and a message for an unknown id).
 * Replaces target_manager_c.h:35 / target_manager_c.cpp:61-65 -> getNumberMeasurements,
 * target_manager.cpp:285-295. */
int  target_manager_get_n_measurements(const target_manager_c *self, const unsigned int id);

/* rt_logger hook of the reference (target_manager_c.h:36 / target_manager_c.cpp:67-71).  The logger is an
 * optional external ROS package there (LOGGER_ON); here, without a log directory the call does nothing, as
 * the reference without LOGGER_ON, and with one (target_manager_set_log_directory in target_batch_c.h, or env
 * TARGET_ESTIMATION_LOG_DIR) it appends the reference's five channels -- measurement, pose, twist,
 * acceleration, covariance (src/target_interface.cpp:32-40) -- of the selected targets to text files. */
void target_manager_log(const target_manager_c *self);

/* Destroy the manager and free its device memory.
 * Replaces target_manager_c.h:37 / target_manager_c.cpp:73-76. */
void target_manager_delete(target_manager_c *self);

#ifdef __cplusplus
}
#endif
#endif
