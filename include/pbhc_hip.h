/* pbhc_hip.h — C ABI of libpbhc_hip.so: the MI355X (gfx950) hot path of the PBHC / KungfuBot
 * humanoidverse motion-tracking agent.
 *
 * The reference (kyungminn/PBHC) is pure Python/PyTorch and has no FFI of its own; its plugin
 * boundary is three Hydra `_target_` classes (simulator / env / algo).  This library is the native
 * layer underneath our drop-in classes for those targets (pbhc_amd/...), and every entry point
 * names the reference code it replaces.  Conventions:
 *   - plain C, no torch types; all tensor arguments are raw DEVICE pointers owned by the caller
 *     (fp32 row-major unless stated; int64 / bool where the reference's tensors are);
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued, never synchronised;
 *   - return 0 on success, a negative PBHC_E* code otherwise; nothing throws across the boundary;
 *   - quaternions are xyzw (Isaac Gym order) except MJCF / extend_config tables, which are wxyz.
 */
#ifndef PBHC_HIP_H
#define PBHC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBHC_ABI_VERSION 11

#define PBHC_OK 0
#define PBHC_EINVAL (-22)   /* bad argument / size over a compile-time maximum */
#define PBHC_ENOMEM (-12)
#define PBHC_EHIP (-5)      /* a HIP runtime call failed; see pbhc_last_error() */

#define PBHC_MAX_BODIES 36   /* bodies incl. extended (hands, head) */
#define PBHC_MAX_DEPTH 12    /* longest root->body chain */
#define PBHC_MAX_DOF 32
#define PBHC_MAX_FEET 2
#define PBHC_MAX_TERMS 32    /* reward terms (vector-reward heads - 1) */
#define PBHC_MAX_IDX 36
#define PBHC_MAX_GROUPS 8    /* observation groups + the history write-back map */
#define PBHC_MAX_QUEUE 8     /* control-delay queue depth */
#define PBHC_NUM_SIGMA 20
#define PBHC_NUM_GLOBALS 128
#define PBHC_MAX_FUTURE 32   /* future reference steps of the general-tracking observations */
#define PBHC_MAX_BODY_Z 8
#define PBHC_NUM_LOG 32

/* ---- skeleton (reference: Humanoid_Batch.__init__/from_mjcf,
 *      humanoidverse/utils/motion_lib/torch_humanoid_batch.py:44-165) ---------------------- */
typedef struct PbhcSkeleton {
  int32_t num_bodies;      /* B  : real bodies                                   */
  int32_t num_bodies_ext;  /* Bx : B + extended bodies                           */
  int32_t num_dof;         /* D  : dof d drives body d+1 (one hinge per body)    */
  int32_t max_depth;
  int32_t parent[PBHC_MAX_BODIES];
  int32_t depth[PBHC_MAX_BODIES];
  float offset[PBHC_MAX_BODIES][3];
  float local_rot_wxyz[PBHC_MAX_BODIES][4];
  float dof_axis[PBHC_MAX_DOF][3];
  /* chain[b][0..chain_len[b]-1]: the non-root bodies on the path root -> b (for an extended body: root -> its parent) */
  int32_t chain_len[PBHC_MAX_BODIES];
  int32_t chain[PBHC_MAX_BODIES][PBHC_MAX_DEPTH];
} PbhcSkeleton;

/* ---- reward term ids (reference: `_reward_<name>` in legged_robot_base.py:944-1087 and
 *      envs/motion_tracking/motion_tracking.py:1154-1328) ------------------------------------ */
enum PbhcRewardTerm {
  PBHC_R_TELEOP_CONTACT_MASK = 0,
  PBHC_R_TELEOP_MAX_JOINT_POSITION,
  PBHC_R_TELEOP_BODY_POSITION_EXTEND,
  PBHC_R_TELEOP_VR_3POINT,
  PBHC_R_TELEOP_BODY_POSITION_FEET,
  PBHC_R_TELEOP_BODY_ROTATION_EXTEND,
  PBHC_R_TELEOP_BODY_ANG_VELOCITY_EXTEND,
  PBHC_R_TELEOP_BODY_VELOCITY_EXTEND,
  PBHC_R_TELEOP_JOINT_POSITION,
  PBHC_R_TELEOP_JOINT_VELOCITY,
  PBHC_R_PENALTY_TORQUES,
  PBHC_R_PENALTY_DOF_VEL,
  PBHC_R_PENALTY_DOF_ACC,
  PBHC_R_PENALTY_ACTION_RATE,
  PBHC_R_PENALTY_ORIENTATION,
  PBHC_R_FEET_AIR_TIME,
  PBHC_R_PENALTY_FEET_CONTACT_FORCES,
  PBHC_R_PENALTY_STUMBLE,
  PBHC_R_PENALTY_SLIPPAGE,
  PBHC_R_FOOT_SLIP_PENALTY,
  PBHC_R_LIMITS_DOF_POS,
  PBHC_R_LIMITS_DOF_VEL,
  PBHC_R_LIMITS_TORQUE,
  PBHC_R_COLLISION,
  PBHC_R_ALIVE,
  /* general tracking (reference: envs/motion_tracking/general_tracking.py:1088-1279) */
  PBHC_R_TELEOP_CONTACT_MASK_V2,
  PBHC_R_TELEOP_KEY_BODY_POSITION,
  PBHC_R_TELEOP_ANCHOR_BODY_POSITION,
  PBHC_R_TELEOP_ANCHOR_BODY_ROTATION,
  PBHC_R_LOCAL_KEY_BODY_POSITION,
  PBHC_R_LOCAL_KEY_BODY_ROTATION,
  PBHC_R_KEY_BODY_VELOCITY,
  PBHC_R_KEY_BODY_ANG_VELOCITY,
  PBHC_R_TELEOP_ROOT_VEL,
  PBHC_R_TELEOP_ROOT_POSE,
  /* foot orientation terms of the base env (legged_robot_base.py:1030-1075; no shipped yaml weights them) */
  PBHC_R_FEET_HEADING_ALIGNMENT,
  PBHC_R_FEET_HEADING_ALIGNMENT_CONTACT,
  PBHC_R_PENALTY_FEET_ORI,
  PBHC_R_PENALTY_FEET_ORI_CONTACT,
  PBHC_R_NUM_TERMS
};

/* ---- tracking-sigma slots (reference: rewards.reward_tracking_sigma keys) ------------------ */
enum PbhcSigma {
  PBHC_S_MAX_JOINT_POS = 0, PBHC_S_UPPER_BODY_POS, PBHC_S_LOWER_BODY_POS, PBHC_S_VR_3POINT_POS,
  PBHC_S_FEET_POS, PBHC_S_BODY_ROT, PBHC_S_BODY_VEL, PBHC_S_BODY_ANG_VEL, PBHC_S_JOINT_POS, PBHC_S_JOINT_VEL,
  /* general tracking */
  PBHC_S_KEY_BODY_POS, PBHC_S_ANCHOR_BODY_POS, PBHC_S_ANCHOR_BODY_ROT, PBHC_S_LOCAL_KEY_BODY_POS, PBHC_S_LOCAL_KEY_BODY_ROT,
  PBHC_S_KEY_BODY_VEL, PBHC_S_KEY_BODY_ANG_VEL, PBHC_S_ROOT_VEL, PBHC_S_ROOT_POSE
};

/* ---- feature ids: the per-env scalars/vectors an observation key can read
 *      (reference getters `_get_obs_<key>`: legged_robot_base.py:1114-1215,
 *      motion_tracking.py:944-1015).  feat_off[id] is the offset inside the feature row. ------- */
enum PbhcFeature {
  PBHC_F_BASE_LIN_VEL = 0, PBHC_F_BASE_ANG_VEL, PBHC_F_PROJECTED_GRAVITY, PBHC_F_DOF_POS, PBHC_F_DOF_VEL,
  PBHC_F_ACTIONS, PBHC_F_REF_MOTION_PHASE, PBHC_F_DIF_LOCAL_RIGID_BODY_POS, PBHC_F_LOCAL_REF_RIGID_BODY_POS,
  PBHC_F_VR_3POINT_POS, PBHC_F_DR_BASE_COM, PBHC_F_DR_LINK_MASS, PBHC_F_DR_KP, PBHC_F_DR_KD, PBHC_F_DR_FRICTION,
  PBHC_F_DR_CTRL_DELAY, PBHC_F_RELYAW, PBHC_F_BASE_POS_Z, PBHC_F_DIF_JOINT_ANGLES, PBHC_F_DIF_JOINT_VELOCITIES,
  PBHC_F_LOCAL_REF_RIGID_BODY_VEL, PBHC_F_GLOBAL_REF_RIGID_BODY_VEL, PBHC_F_HISTORY, PBHC_F_ZERO,
  /* general tracking (getters general_tracking.py:821-954): per-body tables [Bx,3] / [Bx,6], the observation maps pick the key bodies;
   * future targets [S,dim] (general_tracking.py:500-565) */
  PBHC_F_ROLL_PITCH, PBHC_F_CONTACT_MASK, PBHC_F_DR_BASE_MASS, PBHC_F_LOCAL_BODY_POS, PBHC_F_LOCAL_BODY_ROT, PBHC_F_ANCHOR_REF_POS,
  PBHC_F_ANCHOR_REF_ROT, PBHC_F_DIF_ROOT_VELOCITY, PBHC_F_DIF_ROOT_ROT, PBHC_F_DIF_ROOT_HEIGHT, PBHC_F_REF_CONTACT_MASK,
  PBHC_F_FUT_ROOT_HEIGHT, PBHC_F_FUT_ROLL_PITCH, PBHC_F_FUT_BASE_LIN_VEL, PBHC_F_FUT_BASE_ANG_VEL, PBHC_F_FUT_DOF_POS, PBHC_F_FUT_LOCAL_KEY_POS,
  PBHC_F_NUM
};

/* ---- device-resident globals (what the reference keeps as Python floats on the env object:
 *      adaptive sigma + EMA motion_tracking.py:1030-1048, penalty curriculum
 *      legged_robot_base.py:882-900, average_episode_length :875-879, motion-far threshold
 *      motion_tracking.py:309-317).  double[PBHC_NUM_GLOBALS]. -------------------------------- */
enum PbhcGlobal {
  PBHC_G_SIGMA = 0,                       /* [PBHC_NUM_SIGMA] */
  PBHC_G_EMA = 20,                        /* [PBHC_NUM_SIGMA] */
  PBHC_G_PENALTY_SCALE = 40,
  PBHC_G_AVG_EP_LEN = 41,
  PBHC_G_MOTION_FAR_THR = 42,
  PBHC_G_SOFT_POS_VAL = 43,
  PBHC_G_SOFT_VEL_VAL = 44,
  PBHC_G_SOFT_TAU_VAL = 45,
  PBHC_G_STEP_COUNTER = 46,
  PBHC_G_NOISE_CURRICULUM = 47,
  PBHC_G_LOG = 64                         /* [PBHC_NUM_LOG] per-step log means, see PbhcLog */
};

enum PbhcLog {
  PBHC_L_UPPER_BODY_DIFF_NORM = 0, PBHC_L_LOWER_BODY_DIFF_NORM, PBHC_L_VR_3POINT_DIFF_NORM, PBHC_L_JOINT_POS_DIFF_NORM,
  PBHC_L_ACTION_CLIP_FRAC, PBHC_L_RESET_FRAC, PBHC_L_TERM_GRAVITY, PBHC_L_TERM_MOTION_FAR, PBHC_L_TERM_TIME_OUT,
  PBHC_L_TERM_MOTION_END, PBHC_L_END_TIME_RATIO, PBHC_L_END_TIME_RATIO_STD, PBHC_L_NUM_RESETS, PBHC_L_REW_MEAN,
  PBHC_L_KEY_BODY_DIFF_NORM, PBHC_L_LOCAL_UPPER_BODY_DIFF_NORM, PBHC_L_LOCAL_LOWER_BODY_DIFF_NORM, PBHC_L_LOCAL_VR_3POINT_DIFF_NORM,
  PBHC_L_LOCAL_KEY_BODY_DIFF_NORM, PBHC_L_TERM_REF_POS_Z, PBHC_L_TERM_REF_ORI, PBHC_L_TERM_BODY_Z,
  PBHC_L_TERM_CONTACT, PBHC_L_TERM_LOW_HEIGHT,
  PBHC_L_TERM_DOF_POS_LIMIT, PBHC_L_TERM_DOF_VEL_LIMIT, PBHC_L_TERM_TORQUE_LIMIT,
  PBHC_L_NUM
};

/* ---- one output map: out[n][j] = clip((feat[src[j]] + U(-1,1)*noise[j]) * scale[j]) --------
 * (reference: helpers.parse_observation helpers.py:128-152 + sorted-key concat
 *  legged_robot_base.py:787-793 + clip :326-328; group PBHC_MAX_GROUPS-1 is conventionally the
 *  history write-back, history_handler.py:40-44, not clipped) */
/* The same map as RUNS: `len` consecutive output elements dst.. read `len` consecutive feature words src.. with one scale and one noise
 * scale (an observation key is a run; adjacent keys with equal scales merge).  The config-specialised step kernel (pbhc_env_step_spec.hip)
 * unrolls the run list into straight-line code — per element one LDS read at an immediate offset, the scale as a literal, one store —
 * where the generic kernel walks the per-element map. */
#define PBHC_MAX_RUNS 96
typedef struct PbhcObsRun {
  int32_t dst, src, len;
  int32_t late;              /* 1: reads a post-reset feature (written after the reset phase for a terminated env) */
  float scale, noise;
} PbhcObsRun;

typedef struct PbhcOutMap {
  int32_t dim;               /* number of elements this map writes */
  int32_t clip;              /* 1: clip to +-clip_observations */
  int32_t pitch;             /* floats per env row of the output tensor */
  int32_t role;              /* which half of a k_env_step workgroup writes this row: 0 = the dynamics waves (after their reward phase),
                              * 1 = the reference / observation waves; balanced by row width on the host */
  const int32_t* dst;        /* device [dim] position inside the row, or NULL = identity */
  const int32_t* src;        /* device [dim] index into the feature row */
  const float* scale;        /* device [dim] */
  const float* noise;        /* device [dim] noise scale (0 = none) */
  /* compact form (PbhcEnvConfig.map_image): word offset and size of this group's block
   * [seg_scale 16 floats][seg_noise 16 floats][nn_early][nn_late][n_early][n_late][u16 pair index x (n_early + n_late), padded to a word]
   * [noisy entries: j | word[j] << 16, early then late][u16 word[j] x dim, padded to a pair], word[j] = src[j] | seg[j] << 12.
   * A row is written in element pairs (2p, 2p+1): "late" pairs read a post-reset feature, a pair holding a noisy element is in neither
   * list (both of its elements are noise entries) */
  int32_t lds_off;
  int32_t map_words;
  int32_t num_runs;          /* runs[0..num_runs) cover the row exactly once; -1: too many runs for the table (the specialised kernel then
                              * walks the per-element map like the generic one) */
  PbhcObsRun runs[PBHC_MAX_RUNS];
} PbhcOutMap;
#define PBHC_MAX_SEGS 16

/* ---- static configuration of the env (tracking_mode 0: LeggedRobotMotionTracking, 1: LeggedRobotGeneralTracking) -----
 * Filled by the host from the reference's YAML config tree (same keys); copied to the device by
 * pbhc_env_create. */
typedef struct PbhcEnvConfig {
  int32_t abi_version;
  int32_t num_envs;
  PbhcSkeleton skel;
  /* extended bodies: skel entries B..Bx-1 (parent, offset, local_rot) */
  /* timing (base_task.py:37-39) */
  float dt;
  float max_episode_length;
  /* control (legged_robot_base.py:795-838): control_type 0 "P" (position targets), 1 "V" (velocity targets: kp (a - qd) - kd (qd - last_qd) / sim_dt),
   * 2 "T" (scaled actions are the torques), :809-817; sim_dt = 1 / simulator fps (base_task.py:34) */
  int32_t control_type;
  float sim_dt;
  int32_t foot_ori_terms;                    /* 1: a feet_heading_alignment* / penalty_feet_ori* term is configured (per-foot heading and tilt are computed) */
  float p_gains[PBHC_MAX_DOF], d_gains[PBHC_MAX_DOF], action_scale[PBHC_MAX_DOF], default_dof_pos[PBHC_MAX_DOF];
  float torque_limits[PBHC_MAX_DOF], dof_vel_limits[PBHC_MAX_DOF];
  float hard_dof_pos_limits[PBHC_MAX_DOF][2], soft_dof_pos_limits[PBHC_MAX_DOF][2];
  float action_clip_value;
  int32_t clip_torques, randomize_torque_rfi, use_rao, randomize_ctrl_delay, queue_len;
  float rfi_lim;
  /* domain randomisation ranges used on reset (legged_robot_base.py:599-635) */
  int32_t randomize_pd_gain, randomize_rfi_lim;
  /* domain_rand.randomize_default_dof_pos (legged_robot_base.py:632-635): at every reset default_dof_pos[env] = raw default + U(dof_pos_range);
   * needs PbhcStepIO.default_dof_pos (the per-env defaults, read by the torques and the dof_pos observation) */
  int32_t randomize_default_dof_pos;
  float dof_pos_range[2];
  float kp_range[2], kd_range[2], rfi_lim_range[2], rao_lim;
  int32_t ctrl_delay_range[2];
  /* body index sets (base_task.py:169-205, motion_tracking.py:203-232) */
  int32_t num_feet, feet[PBHC_MAX_FEET];
  int32_t num_penalised, penalised[PBHC_MAX_IDX];
  int32_t num_upper, upper[PBHC_MAX_IDX];
  int32_t num_lower, lower[PBHC_MAX_IDX];
  int32_t num_track, track[PBHC_MAX_IDX];
  int32_t body_flags[PBHC_MAX_BODIES];       /* bit0 upper, bit1 lower, bit2 tracked(vr 3-point), bit3 foot, bit4 key body, bit5 body_z list */
  int32_t track_slot[PBHC_MAX_BODIES];       /* position of the body inside `track`, or -1 */
  /* termination (legged_robot_base.py:408-489, motion_tracking.py:330-357) */
  int32_t terminate_by_gravity, terminate_when_motion_far, terminate_when_motion_end, motion_far_curriculum;
  float termination_gravity;
  float motion_far_degree, motion_far_down, motion_far_up, motion_far_min, motion_far_max;
  /* rewards (legged_robot_base.py:167-233,715-761) */
  int32_t num_terms;                         /* columns 0..num_terms-1; rew_buf has num_terms+1 columns if use_vec_reward */
  int32_t use_vec_reward, num_rew_cols;
  int32_t term_id[PBHC_MAX_TERMS];
  float term_scale[PBHC_MAX_TERMS];          /* reward_scales[name] * dt */
  int32_t term_penalty[PBHC_MAX_TERMS];      /* in reward_penalty_reward_names and curriculum on */
  int32_t term_sum_col[PBHC_MAX_TERMS];      /* column in episode_sums */
  int32_t term_src[PBHC_MAX_TERMS];          /* filled by pbhc_env_create (kernel-internal slot of the term's raw value, or -1) */
  int32_t sum_col_term[32];                  /* filled by pbhc_env_create: the term that accumulates into episode_sums column i, or -1 */
  int32_t has_termination, termination_sum_col, only_positive_rewards, num_sum_cols;
  float termination_scale;
  float body_pos_lower_weight, body_pos_upper_weight, desired_feet_air_time, max_contact_force;
  int32_t adaptive_sigma;                    /* rewards.adaptive_tracking_sigma.enable */
  int32_t adaptive_type;                     /* 0 "origin" (min(ema, sigma)), 1 "mean" (v1: (min+ema)/2 motion_tracking.py:1040-1045; v2: ema
                                                general_tracking.py:988-989), 2 "scale" (min(ema*scale, sigma)) */
  float adaptive_scale;
  int32_t sigma_active[PBHC_NUM_SIGMA];      /* 1 if a configured reward term updates this sigma */
  float adaptive_alpha;
  int32_t penalty_curriculum;
  float penalty_degree, penalty_down, penalty_up, penalty_min, penalty_max;
  /* obs.add_noise_currculum (legged_robot_base.py:591-592,1117-1126): PBHC_G_NOISE_CURRICULUM scales every observation noise amplitude and
   * moves by (1 -/+ degree) when the average episode length is below `noise_down` / above `noise_up` (the reference reads the latter from
   * rewards.reward_penalty_level_up_threshold), clipped to [noise_min, noise_max] */
  int32_t noise_curriculum;
  float noise_degree, noise_down, noise_up, noise_min, noise_max;
  int32_t num_compute_average_epl;
  int32_t soft_pos_curriculum, soft_vel_curriculum, soft_tau_curriculum;
  /* rewards.reward_limit.reward_limits_curriculum (legged_robot_base.py:902-939), [0] dof_pos, [1] dof_vel, [2] torque: at every step that
   * resets an env the value PBHC_G_SOFT_{POS,VEL,TAU}_VAL moves by (1 +/- degree) when the average episode length is below `down` / above `up`
   * — the limits WIDEN while episodes are short — and is clipped to [min, max] (the shipped yamls have min == max: a constant) */
  float soft_cur_degree[3], soft_cur_down[3], soft_cur_up[3], soft_cur_min[3], soft_cur_max[3];
  float soft_dof_vel_limit, soft_torque_limit;
  float max_episode_length_s;
  /* observations */
  float clip_observations;
  int32_t feat_off[PBHC_F_NUM];              /* features no observation reads share one scratch region at the end of the row */
  int32_t feat_dim;
  int32_t hist_dim;                          /* floats of history state per env */
  int32_t num_groups;
  PbhcOutMap groups[PBHC_MAX_GROUPS];
  /* compact observation maps, staged in LDS once per workgroup when they fit without costing occupancy (requires identity dst,
   * feat_dim <= 4096 and at most PBHC_MAX_SEGS distinct (scale, noise) pairs per group): the concatenated group blocks */
  int32_t map_lds_words;                     /* 0: none */
  int32_t pad3_;
  const uint32_t* map_image;                 /* device [map_lds_words] */
  int32_t has_contact_mask;
  float ref_init_yaw;
  int32_t dr_link_mass_dim;
  /* general tracking (general_tracking.py:86-98,226-262,500-507) */
  int32_t tracking_mode;
  int32_t num_key, key[PBHC_MAX_IDX];        /* robot.key_bodies as indices into the extended body list */
  int32_t key_slot[PBHC_MAX_BODIES];         /* position of the body inside `key`, or -1 */
  int32_t anchor_index;                      /* find_rigid_body_indice(anchor_link) + 1, sic */
  int32_t future_num_steps, future_steps[PBHC_MAX_FUTURE];   /* linspace(1, future_max_steps, future_num_steps) as long */
  int32_t terminate_by_ref_pos_z, terminate_by_ref_ori, terminate_by_body_z;
  float ref_pos_z_threshold, ref_ori_threshold, body_z_threshold;
  /* legged_robot_base.py:434-444: any |contact force| > 1 N on robot.terminate_after_contacts_on bodies; root height below a minimum */
  int32_t terminate_by_contact, terminate_by_low_height;
  int32_t num_term_contact, term_contact[PBHC_MAX_IDX];
  float termination_min_base_height;
  /* legged_robot_base.py:449-479: with probability term_close_prob[g] PER STEP (one draw for all envs) an env terminates when any joint is beyond
   * its termination position limit (isaacgym.py:387-388), |dof_vel| > dof_vel_limits * term_close_vel_scale, |torque| > torque_limits * term_close_tau_scale */
  int32_t terminate_close_pos, terminate_close_vel, terminate_close_tau;
  float term_close_prob[3];
  float dof_pos_limits_termination[PBHC_MAX_DOF][2];
  float term_close_vel_scale, term_close_tau_scale;
  int32_t pad2_;
  uint64_t seed;
} PbhcEnvConfig;

/* ---- reference-motion table (reference: MotionLibBase.load_motions / get_motion_state,
 *      motion_lib_base.py:123-259,261-391).  One packed row per frame:
 *      [dof_pos D | dof_vel D | contact 2 | pos Bx*3 | rot Bx*4 | vel Bx*3 | ang Bx*3]. ---------- */
typedef struct PbhcMotionTable {
  const float* frames;          /* device [total_frames, row] */
  int32_t row;                  /* 2D + 2 + 13 Bx */
  int32_t num_motions;
  const int32_t* length_starts; /* device [M] first row of each clip   */
  const int32_t* num_frames;    /* device [M]                           */
  const float* motion_dt;       /* device [M] 1/fps                     */
  const float* motion_len;      /* device [M] (F-1)/fps                 */
  /* when num_motions == 1 the clip's meta also travels by value (saves a dependent load in the step kernel) */
  int32_t single_num_frames;
  float single_dt, single_len;
} PbhcMotionTable;

/* ---- per-step tensors of the fused env step ------------------------------------------------ */
typedef struct PbhcStepIO {
  /* inputs */
  const float* actions_in;        /* [N,D]   policy actions                                    */
  /* replay window of the sim-stub: frame k = frame_cursor[0] % num_frames lands this step; the cursor lives on the device and is
   * advanced by the step itself (pointer bump without a host round trip, so a rollout can be captured in a hipGraph). */
  const float* frame_root;        /* [T,N,13]                                                  */
  const float* frame_dof_pos;     /* [T,N,D]                                                   */
  const float* frame_dof_vel;     /* [T,N,D]                                                   */
  const float* frame_contact;     /* [T,N,B,3] net contact forces                              */
  int32_t* frame_cursor;          /* device int32[1]                                           */
  int32_t num_frames;             /* T                                                         */
  int32_t frame_index;            /* >= 0: the host names the frame (saves the kernel a dependent load; the device cursor is still
                                     advanced); < 0: read the device cursor (hipGraph replay)  */
  /* optional injected random draws (NULL -> in-kernel Philox) */
  const float* u_rfi;             /* [N,D] uniforms of the torque RFI noise                    */
  const float* ovr_start_time;    /* [N]   values consumed by resetting envs                   */
  const float* ovr_kp; const float* ovr_kd; const float* ovr_rfi_lim; const float* ovr_rao; /* [N,D] */
  const float* ovr_dof_pos_bias;                                                          /* [N,D] the U(dof_pos_range) draw of randomize_default_dof_pos */
  const int64_t* ovr_delay;       /* [N]                                                       */
  const float* ovr_gate_u;        /* [3]   the per-step uniforms of the terminate_when_close_to_* gates (pos, vel, torque) */
  /* simulator-surface state (reference names; simulator/isaacgym/isaacgym.py:574-618) */
  float* root_states;             /* [N,13] */
  float* dof_state;               /* [N,D,2] (pos, vel) */
  float* rigid_body_state;        /* [N,B,13] pos3 rot4 vel3 ang3, may be NULL: nothing on the training path reads it — the replay stub hands NULL and
                                   * re-derives it on first access (pbhc_sim_fk of the same replay frame: the same arithmetic) */
  float* contact_forces;          /* [N,B,3], may be NULL (a copy of the replay frame's contact forces) */
  /* env state (reference names; legged_robot_base.py:39-131, motion_tracking.py:251-263) */
  float* actions; float* last_actions; float* actions_after_delay; float* action_queue; /* [N,D] x3, [N,Q,D] */
  float* last_dof_pos; float* last_dof_vel; float* torques;                              /* [N,D] */
  float* feet_air_time; float* contacts; float* contacts_filt; float* last_contacts; float* last_contacts_filt; /* [N,F] */
  float* kp_scale; float* kd_scale; float* rfi_lim_scale; float* rao_scale;              /* [N,D] */
  float* default_dof_pos;                                                                 /* [N,D] or NULL: per-env default joint angles (randomize_default_dof_pos); NULL -> PbhcEnvConfig.default_dof_pos */
  float* motion_start_times; float* motion_len; float* end_time_ratio_buf;               /* [N] */
  float* episode_sums;            /* [N,num_sum_cols] */
  float* hist;                    /* [N,hist_dim] */
  int64_t* episode_length_buf; int64_t* last_episode_length_buf; int64_t* reset_buf; int64_t* action_delay_idx; /* [N] */
  const int64_t* motion_ids;      /* [N] slot -> clip */
  uint8_t* time_out_buf;          /* [N] bool */
  const float* env_origins;       /* [N,3] */
  const float* dr_base_com; const float* dr_link_mass; const float* dr_friction;        /* [N,3] [N,L] [N,1] */
  const float* dr_base_mass;      /* [N,1] (general tracking priv_obs), may be NULL -> 1.0 */
  /* outputs */
  float* obs[PBHC_MAX_GROUPS];    /* [N,dim_g]; group num_groups-1 may alias `hist` semantics (see PbhcOutMap) */
  float* rew_buf;                 /* [N,num_rew_cols] */
  float* ref_body_pos_extend;     /* [N,Bx,3] may be NULL: extras["ref_body_pos_extend"] is read by evaluation callbacks only — the env hands NULL and */
  float* ref_body_rot_extend;     /* [N,Bx,4] may be NULL  re-derives both on first access from ref_time_out (pbhc_motion_state: the same lerp / slerp) */
  float* ref_time_out;            /* [N] may be NULL: the motion time of this step's reference lookup, (episode_length + 1) dt + start BEFORE a reset */
  float* episode_rew_out;         /* [N,num_sum_cols] episode_sums / max_episode_length_s of envs reset this step (else unchanged), may be NULL */
  /* Data-parallel runs: NULL -> the step finalizes its batch statistics itself.  Non-NULL -> device double[PBHC_NUM_TOTALS]: the step
   * only writes this shard's batch sums there (and advances the RNG counter / frame cursor); the caller sums them over ranks and calls
   * pbhc_env_finalize, so adaptive sigma, average episode length, the curricula and the logged means are those of ONE batch of
   * all ranks' envs (the reference is single-process: motion_tracking.py:1030-1048, legged_robot_base.py:875-900). */
  double* totals_out;
  /* row pitches in floats (0 = dense).  Rows padded to a multiple of 32 floats start on 128-B lines: no cache line is shared by two envs,
   * so no line is written twice by different workgroups. */
  int32_t obs_pitch[PBHC_MAX_GROUPS];
  int32_t hist_pitch;
  /* 1: re-draw the episodic domain randomisation (gain / torque-noise scales, control delay + action queue, default joint angles) of EVERY
   * env in this step, after its torques and before its observations — `_update_tasks_callback` with domain_rand.reinit_epis_rand > 0
   * (legged_robot_base.py:390-395 -> _episodic_domain_randomization(all env ids) :599-635).  Same draws as a reset's (Philox streams / ovr_*). */
  int32_t redraw_all;
  /* written by pbhc_env_step_launch itself (callers leave it 0): 1 when every observation row may be stored 16 bytes per lane — obs[g]
   * 16-byte aligned, its pitch a multiple of 4 floats and >= the row width rounded up to 4 (the env-owned and the rollout-buffer rows are:
   * padded to 128-byte lines); the words between a row's width and that bound are then written (zeros). */
  int32_t obs_wide;
} PbhcStepIO;

typedef struct PbhcEnv PbhcEnv;   /* opaque */

int pbhc_abi_version(void);
const char* pbhc_last_error(void);
int pbhc_sizeof_env_config(void);
int pbhc_sizeof_step_io(void);

/* Load-time FK + filtered velocities of one clip -> packed frame rows.
 * Replaces Humanoid_Batch.fk_batch / _compute_velocity / _compute_angular_velocity
 * (torch_humanoid_batch.py:168-290).  pose_aa [F,Bx,3], trans [F,3], contact [F,2] or NULL (device);
 * out_rows [F,row] device; scratch >= F*Bx*14 floats device. */
int pbhc_motion_build(const PbhcSkeleton* skel, const float* pose_aa, const float* trans, const float* contact,
                      int num_frames, float dt, float* out_rows, float* scratch, void* stream);

/* The same for a whole library in ONE launch set (the reference runs the FK once per env slot in a Python loop over clips,
 * motion_lib_base.py:403-472): the clips' frames concatenated — pose_aa [total_frames,Bx,3], trans [total_frames,3], contact [total_frames,2]
 * or NULL — with frame_clip [total_frames] (clip of every frame), clip_start [num_clips+1] (first frame of every clip, then the total) and
 * clip_dt [num_clips] (1 / fps), all device int32 / float.  Velocities and the Gaussian filter stop at clip boundaries.
 * out_rows [total_frames,row]; scratch >= total_frames*Bx*14 floats. */
int pbhc_motion_build_batch(const PbhcSkeleton* skel, const float* pose_aa, const float* trans, const float* contact, int total_frames, int num_clips,
                            const int32_t* frame_clip, const int32_t* clip_start, const float* clip_dt, float* out_rows, float* scratch, void* stream);

/* Phase lookup + lerp/slerp.  Replaces MotionLibBase.get_motion_state (motion_lib_base.py:123-259).
 * ids [N] int64, times [N], offset [N,3] or NULL -> out [N,row] packed like a frame row
 * (positions include the offset). */
int pbhc_motion_state(const PbhcMotionTable* tbl, int num_bodies_ext, int num_dof, const int64_t* ids, const float* times,
                      const float* offset, int n, float* out, void* stream);

/* Rigid-body pose + twist from (root state, q, q-dot): what Isaac Gym's rigid-body state tensor
 * supplied in the reference (isaacgym.py:574-605).  out [N,B,13]. */
int pbhc_sim_fk(const PbhcSkeleton* skel, const float* root_states, const float* dof_pos, const float* dof_vel,
                int dof_stride, int n, float* out_body_state, void* stream);
/* Test-only: the same chain for every body INCLUDING the extended ones (motion_tracking.py:619-643), out [N,Bx,13], by method 0 — the walk
 * pbhc_sim_fk runs — or 1 — the pointer-jumping form the fused step runs for robots of <= 32 bodies (csrc/pbhc_env_step.h: fk_jump_wave).
 * dof_pos / dof_vel [N,D] dense. */
int pbhc_debug_fk(const PbhcSkeleton* skel, const float* root_states, const float* dof_pos, const float* dof_vel, int n, int method,
                  float* out_body_state_ext, void* stream);

/* Env object: owns the device copy of the config, the globals and the reduction scratch. */
/* `globals`: caller-owned device double[PBHC_NUM_GLOBALS] (see enum PbhcGlobal), initialised by the caller. */
int pbhc_env_create(const PbhcEnvConfig* cfg, const PbhcMotionTable* tbl, double* globals, PbhcEnv** out);
void pbhc_env_destroy(PbhcEnv* env);

/* Config-specialised step kernel (new; nothing of the reference to match — its counterpart there is TorchScript specialising the
 * rotation helpers per call site, isaac_utils/rotations.py `@torch.jit.script`).  k_env_step reads ~100 config scalars; compiled with the
 * env's config as compile-time constants (csrc/pbhc_env_step_spec.hip, -DPBHC_STATIC_CFG) the same source is 40 % shorter and ~19 % faster.
 *   pbhc_env_get_config          the config as pbhc_env_create finalised it: the text a specialised build is generated from
 *   pbhc_env_attach_specialised  load `so_path` (a build of pbhc_env_step_spec.hip) and use its kernel for this env's steps; refused with
 *                                PBHC_EINVAL unless the config baked into the object equals this env's member by member (run-time members —
 *                                pointers, num_envs, seed, ref_init_yaw — excepted: the kernel reads those from the env).  NULL detaches.
 *   pbhc_env_is_specialised      1 while a specialised kernel is attached
 *   pbhc_env_config_finalize     the same finalisation without an env or a device (validation + derived members): lets a build step
 *                                generate the specialised kernels of known configs ahead of time */
int pbhc_env_get_config(PbhcEnv* env, PbhcEnvConfig* out);
int pbhc_env_config_finalize(const PbhcEnvConfig* cfg, PbhcEnvConfig* out);
/* Dynamic LDS (bytes) of one k_env_step workgroup — 4 envs: state, bodies, reference frame, feature row, skeleton constants, compact
 * observation maps — for this config; < 0: -(error code).  160 KB per CU / this = resident workgroups per CU (no reference counterpart). */
int pbhc_env_config_lds_bytes(const PbhcEnvConfig* cfg);
int pbhc_env_attach_specialised(PbhcEnv* env, const char* so_path);
int pbhc_env_is_specialised(PbhcEnv* env);
/* One LeggedRobotBase.step (legged_robot_base.py:239-338) for all envs: 2 launches (step + finalize). */
int pbhc_env_step(PbhcEnv* env, const PbhcStepIO* io, void* stream);
/* The same step as its two launches, for a caller that keeps the one-workgroup reduction off its critical chain (the rollout of
 * mh_ppo.py:270-342 is step -> policy forward -> step: only the fused launch has to precede the policy forward).  `launch` = the fused
 * per-env kernel on `stream`; `finish` = the reduction of its partial sums into the globals (or into io->totals_out) on a stream of the
 * caller's choice, ordered by the caller after the launch and before the next launch / pbhc_policy_sample (which reads the step counter). */
int pbhc_env_step_launch(PbhcEnv* env, const PbhcStepIO* io, void* stream);
int pbhc_env_step_finish(PbhcEnv* env, const PbhcStepIO* io, void* stream);
/* Second half of a step launched with io->totals_out: `totals` = the element-wise sum over ranks of every shard's totals_out,
 * `num_envs_total` = the number of envs of all ranks.  Must run before the next pbhc_env_step of this env. */
#define PBHC_NUM_TOTALS 64
int pbhc_env_finalize(PbhcEnv* env, const double* totals, double num_envs_total, void* stream);

/* Measurement aid: when enabled, k_env_step of each following pbhc_env_step carries a pair of HIP
 * events attached to its dispatch (kernel begin / end) on the launch stream (ring of PBHC_PROFILE_RING pairs).  pbhc_env_profile_read synchronises on the
 * recorded events and returns the most recent durations in milliseconds (oldest first). */
#define PBHC_PROFILE_RING 512
int pbhc_env_profile(PbhcEnv* env, int enable);
int pbhc_env_profile_read(PbhcEnv* env, float* ms_out, int max_count, int* count);
/* What the dispatch-attached event pair reads BEYOND a kernel's execution (queue-side start / stop handling): measured with a one-wave kernel
 * that spins for exactly 20 us of the 100 MHz wall clock (median of 33 launches minus 0.020 ms).  Subtract it from pbhc_env_profile_read's values
 * to compare with a profiler's kernel durations. */
int pbhc_env_profile_overhead(PbhcEnv* env, void* stream, float* overhead_ms);

/* Rollout-side fusions of MHPPO._rollout_step (mh_ppo.py:270-342).
 * pbhc_policy_sample: a ~ Normal(mu, std) (Philox keyed by seed / counter[0] / env), log-prob, and the writes of one rollout-buffer
 * step: actions, action_mean, action_sigma [N,A], actions_log_prob [N], values [N,R] (copied from `value`; both may be NULL when the
 * critic runs on another stream and stores its own output).
 * counter: device double, read (not advanced) — pass the env globals' step counter. */
int pbhc_policy_sample(const float* mu, const float* std, const float* value, int N, int A, int R, uint64_t seed, const double* counter,
                       float* actions, float* action_mean, float* action_sigma, float* logp, float* values_out, void* stream);
/* pbhc_rollout_post: rewards_stored = rew + gamma * values * time_out (bootstrap, :300-305), dones = reset_buf != 0, and the
 * episode book-keeping (:311-323) kept on the device: cur_reward_sum/cur_episode_length [N], ep_stats double[3] +=
 * (sum of finished returns, sum of finished lengths, count).
 * pbhc_rollout_post2: the same with `values` NULL allowed — rewards_stored = rew, the caller adds the bootstrap once the values exist (the
 * critic evaluated over all steps' slabs at once) — and the step's time-out flags copied to time_outs_out [N] (may be NULL) for that. */
int pbhc_rollout_post(const float* rew, const float* values, const int64_t* reset_buf, const uint8_t* time_outs, int N, int R, float gamma,
                      float* rewards_out, uint8_t* dones_out, float* cur_reward_sum, float* cur_episode_length, double* ep_stats, void* stream);
int pbhc_rollout_post2(const float* rew, const float* values, const int64_t* reset_buf, const uint8_t* time_outs, int N, int R, float gamma,
                       float* rewards_out, uint8_t* dones_out, float* cur_reward_sum, float* cur_episode_length, double* ep_stats, uint8_t* time_outs_out,
                       void* stream);

/* GAE + returns + normalised advantages.  Replaces MHPPO._compute_returns (mh_ppo.py:348-395).
 * rewards/values/returns [T,N,R], dones [T,N] bool, last_values [N,R], advantages [T,N].
 * stats: device double[4] scratch. */
int pbhc_gae(const float* rewards, const float* values, const uint8_t* dones, const float* last_values, int T, int N, int R,
             float gamma, float lam, float* returns, float* advantages, double* stats, void* stream);

/* Fused forward + backward of the MHPPO losses (mh_ppo.py:433-480,509-511) on the network outputs of one minibatch.
 * mu [B,A], std [A] (the actor's `std` parameter), value [B,R]; batch tensors as the reference's storage keys
 * (actions [B,A], old_logp [B], old_mu/old_sigma [B,A], adv [B], returns/old_values [B,R]).
 * Out: grad_mu = d(actor_loss)/d(mu) [B,A], grad_value = d(critic_loss)/d(value) [B,R], grad_std [A] (surrogate part and the
 * entropy bonus), scalars[4] = {surrogate loss, value loss, entropy, mean KL}.  adapt_lr bit 0: apply the adaptive-KL rule
 * (mh_ppo.py:455-466) to lr[0] (actor) and lr[1] (critic) on the device; bit 1: the KL of ppo_mimic.py:621-628
 * (log(sigma / (old_sigma + 1e-5)) instead of log(sigma / old_sigma + 1e-5)).  `std` is the sigma vector the distribution used
 * (mh_ppo: the parameter; ppo_mimic: clamp(std, min_sigma, max_sigma), the caller masks grad_std accordingly).
 * scalars_acc (may be NULL): float[4], scalars_acc[i] += scalars[i] — the running sums behind the iteration's mean losses
 * (mh_ppo.py:412-417 `mean_value_loss += ...`) without a launch of their own.  scratch: pbhc_ppo_loss_scratch_floats(B) floats. */
int pbhc_ppo_loss(const float* mu, const float* std, const float* value, const float* actions, const float* old_logp, const float* old_mu,
                  const float* old_sigma, const float* adv, const float* returns, const float* old_values, int B, int A, int R, float clip,
                  float value_coef, float entropy_coef, int use_clipped_value_loss, float desired_kl, int adapt_lr, float* grad_mu, float* grad_value,
                  float* grad_std, float* scalars, float* scalars_acc, float* lr, float* scratch, void* stream);
int pbhc_ppo_loss_scratch_floats(int B);
/* The same learning-rate rule on a KL mean already on the device (data-parallel update: the mean over all ranks, after the all-reduce):
 * lr[i] <- rule(lr[i], *kl_mean) for i < n.  Reference: mh_ppo.py:455-466. */
int pbhc_kl_lr_rule(float* lr, int n, const float* kl_mean, float desired_kl, void* stream);

/* Backward of one MLP activation fused with the bias gradient of the layer below it (what autograd runs as elu_backward /
 * silu_backward + a column sum, agents/modules/modules.py:47-63 under torch.autograd): dz = dy * act'(saved), grad_bias = colsum(dz).
 * act: 0 none (bias gradient of the output layer), 1 ELU from the activation output, 2 SiLU from the pre-activation, 3 ReLU from the
 * output.  dy/saved/dz [B,n] row-major (dz may alias dy), grad_bias [n], scratch >= PBHC_ACT_MAX_BLOCKS * n floats. */
#define PBHC_ACT_MAX_BLOCKS 1024
int pbhc_act_bwd_bias(const float* dy, const float* saved, int B, int n, int act, float* dz, float* grad_bias, float* scratch, void* stream);
/* The same in two halves, so that a whole network's bias gradients are finished by ONE launch: pbhc_act_bwd_partials does the slab pass
 * and leaves per-row-block column sums in `scratch` (returns their count in *num_row_blocks); pbhc_colsum_final sums up to
 * PBHC_MAX_COLSUM_JOBS such scratch areas into their grad_bias vectors (fixed order, deterministic). */
#define PBHC_MAX_COLSUM_JOBS 16
typedef struct PbhcColsumJob {
  const float* part;   /* [num_row_blocks, n] */
  float* out;          /* [n] */
  int32_t num_row_blocks;
  int32_t n;
} PbhcColsumJob;
int pbhc_act_bwd_partials(const float* dy, const float* saved, int B, int n, int act, float* dz, float* scratch, int* num_row_blocks, void* stream);
int pbhc_colsum_final(const PbhcColsumJob* jobs, int num_jobs, void* stream);
/* Backward of a stack's narrow OUTPUT Linear (y = h W^T + b, A = out_features <= 32, K = in_features in {64, 128, 192, 256}) in one pass over the
 * rows — what autograd runs as mm (weight gradient), a column sum (bias gradient), mm (input gradient) and the activation backward of the layer
 * below (agents/modules/modules.py:47-63 under torch.autograd; mh_ppo.py:513-517):
 *   dh[M,K] = (dy[M,A] . w[A,K]) * act'(saved)       saved = the lower layer's activation OUTPUT (ELU, ReLU; NULL: = h) or PRE-activation (SiLU)
 *   part_dw[b, A*K], part_db[b, A], part_cs[b, K]    partial rows b < *num_row_blocks (<= PBHC_ACT_MAX_BLOCKS) of dW = dy^T h, db = colsum(dy) and
 *                                                    colsum(dh); finish each with a pbhc_colsum_final job (n = A*K, A, K). */
int pbhc_linear_out_bwd(const float* dy, const float* h, const float* saved, const float* w, int M, int A, int K, int act, float* dh, float* part_dw,
                        float* part_db, float* part_cs, int* num_row_blocks, void* stream);
/* test / measurement aid.  bits 0-7: 0 = the streaming VALU form of pbhc_linear_out_bwd for every shape (default 1: K = 128 runs on the matrix
 * cores); bits 8-15: 32-row tiles per workgroup of the matrix-core form (0 / 1: one — measured in the update: 28.85 / 28.84 / 29.1 ms per update
 * at 1 / 2 / 3, i.e. fewer partial rows buy nothing) */
void pbhc_debug_out_bwd_variant(int mfma);


/* One Linear of a training-time Linear / activation stack on the fp32 matrix cores with its activation folded into the GEMM epilogue
 * (agents/modules/modules.py:47-63: `nn.Linear` followed by `nn.ELU` / `nn.SiLU` / `nn.ReLU`; replaces torch.addmm + the activation pass):
 *   y[M,N] = act(x[M,K] . w[N,K]^T + bias[N])        act: 0 none, 1 ELU (alpha 1), 2 SiLU, 3 ReLU;  bias may be NULL
 *   pre[M,N] (may be NULL) = x . w^T + bias            the pre-activation, which SiLU's derivative needs
 * x, w row-major, contiguous (rows need 4-byte alignment only).  f32 in, f32 accumulate (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain). */
int pbhc_linear_act_fwd(const float* x, const float* w, const float* bias, float* y, float* pre, int M, int N, int K, int act, void* stream);
/* The stack's last hidden layer AND its narrow output layer in one launch (the tail `Linear(K,128) - act - Linear(128,NO)` of modules.py:47-63's
 * nn.Sequential: 23 action means / 20 value heads): y, pre as above with N = 128, and out[M, NO] = y . w_out[NO,128]^T + b_out[NO] (b_out may be
 * NULL) computed from the tile's activated rows while they are in LDS (32-row tiles hold whole rows) — replaces the library addmm of the
 * output layer (14 / 9 us of launch for 0.14 GFLOP at 24 576 rows).  NO <= 32, K >= 4; out rows contiguous. */
int pbhc_linear_act_fwd_out(const float* x, const float* w, const float* bias, float* y, float* pre, int M, int N, int K, int act, const float* w_out,
                            const float* b_out, int NO, float* out, void* stream);
/* The same with row strides and a batch: batch b computes y[b] = act(x[b] . w^T + bias) with x[b] = x + b * x_batch_stride (rows lda floats
 * apart, lda >= K), y[b] = y + b * y_batch_stride (rows ldc floats apart, ldc >= N), one weight matrix for all — e.g. the L output positions of
 * nn.Conv1d(C -> O, kernel k, stride s) on time-major activations [B, T, C] (encoder_modules.py:60-107): window l is the [B, k*C] matrix at
 * x + l*s*C with lda = T*C, written to out[:, l, :] of [B, L, O] (y + l*O, ldc = L*O).  K >= 4. */
int pbhc_linear_act_fwd_strided(const float* x, int lda, long long x_batch_stride, const float* w, const float* bias, float* y, float* pre, int ldc,
                                long long y_batch_stride, int batches, int M, int N, int K, int act, void* stream);
/* The WHOLE Linear / activation stack of a no-grad forward (the rollout's policy and critic evaluation, `BaseModule.forward`,
 * agents/modules/modules.py:47-63 as called from mh_ppo.py:286-290 / ppo_mimic.py:392-400) in one launch: a workgroup carries 16 rows through
 * every layer, activations stay in LDS.  y[M, dims[L]] = W_L-1 act(... act(W_0 x + b_0) ...) + b_L-1 (no activation after the last layer).
 * x rows ldx floats apart, y rows ldy apart; weights[l] = the PACKED copy (pbhc_mlp_pack, pbhc_mlp_packed_floats(dims[l+1], dims[l]) floats,
 * 16-byte aligned) of nn.Linear.weight [dims[l+1], dims[l]] — the MFMA operand fragments laid out 1 KiB-contiguous per wave-instruction;
 * the caller repacks when the weights change (once per rollout: they are constant over its steps).  biases[l] may be NULL; act as
 * pbhc_linear_act_fwd.  pbhc_mlp_fwd_lds_bytes: the launch's dynamic LDS (<= 160 KB or the call is refused). */
#define PBHC_MLP_MAX_LAYERS 8
/* pbhc_mlp_fwd_sample: the policy's stack with pbhc_policy_sample folded into the last layer's epilogue (mh_ppo.py:286-296: the rollout's
 * act() + log-prob + buffer writes): mu = the stack's output [M, A] (A = dims[L] <= 32), actions = mu + std * z with z from the Philox call
 * pbhc_policy_sample makes — keyed by seed / (uint32)counter[0] + counter_offset / row / column — action_mean = mu, action_sigma = std,
 * logp[M] = the row's Normal log-density.  `counter`: a device double the caller keeps constant over a rollout (a snapshot of the env's step
 * counter at its start) with counter_offset = the step index, so that the launch does not wait for the previous step's reduction. */
typedef struct PbhcMlpSample {
  const float* std;            /* [A] */
  const double* counter;
  uint64_t seed;
  int32_t counter_offset;
  int32_t pad_;
  float* actions;              /* [M, A] */
  float* action_mean;          /* [M, A] */
  float* action_sigma;         /* [M, A] */
  float* logp;                 /* [M] */
} PbhcMlpSample;
int pbhc_mlp_fwd_sample(const float* x, int ldx, const float* const* weights, const float* const* biases, const int* dims, int num_layers, int act, int M,
                        const PbhcMlpSample* sample, void* stream);
/* pbhc_mlp_fwd_cat: the same stack on input rows given as up to PBHC_MLP_MAX_SEGS column segments laid side by side — `module(torch.cat([obs,
 * motion_embedding, latent], -1))` of the general-tracking actor (agent_modules.py:75-84 of the reference) without materialising the concatenation;
 * sum(width) == dims[0], ld[i] >= width[i] (floats).  y may be NULL when `sample` is given (then the sampling epilogue of pbhc_mlp_fwd_sample runs);
 * both may be given.  16-byte loads are used when every segment allows them (16-byte aligned base, ld % 4 == 0, inner widths % 4 == 0). */
#define PBHC_MLP_MAX_SEGS 3
typedef struct PbhcMlpInput {
  const float* x[PBHC_MLP_MAX_SEGS];
  int32_t ld[PBHC_MLP_MAX_SEGS];
  int32_t width[PBHC_MLP_MAX_SEGS];
  int32_t nseg;
} PbhcMlpInput;
int pbhc_mlp_fwd_cat(const PbhcMlpInput* in, const float* const* weights, const float* const* biases, const int* dims, int num_layers, int act, float* y, int ldy,
                     int M, const PbhcMlpSample* sample, void* stream);
/* pbhc_conv_encoder_fwd: the general-tracking `ConvEncoder` (agents/modules/encoder_modules.py:22-107 of the reference: Linear(d -> H) + ReLU
 * per time step, Conv1d(H -> O1, k1, s1) + act, Conv1d(O1 -> O2, k2, s2) + act, Linear(L2 * O2 -> E)) under no_grad in ONE launch, 16 rows per
 * workgroup through all four layers (the rollout: 4 096 rows per control step).  x [M, T * d] with row pitch ldx >= (T - 1) * d + ceil16(d)
 * (a rollout slab's padded rows), y [M, E].  Weights packed by pbhc_mlp_pack from row-major matrices: w1 [H, d]; wc1 [O1, k1 * H] and wc2
 * [O2, k2 * O1] with the kernel tap OUTER (conv.weight.permute(0, 2, 1)); wo [E, L2 * O2] with its columns in (position, channel) order.
 * act: 1 ELU, 2 SiLU, 3 ReLU (the conv layers; layer 1 is ReLU, the output layer linear).  pbhc_conv_encoder_lds_bytes: the launch's LDS need
 * (<= 160 KB; the caller falls back to the per-layer kernels otherwise). */
typedef struct PbhcConvEncoder {
  const float* w1; const float* b1;
  const float* wc1; const float* bc1;
  const float* wc2; const float* bc2;
  const float* wo; const float* bo;
  int32_t T, d, H, O1, k1, s1, O2, k2, s2, E, act;
  int32_t pad_;
} PbhcConvEncoder;
size_t pbhc_conv_encoder_lds_bytes(const PbhcConvEncoder* e);
int pbhc_conv_encoder_fwd(const float* x, int ldx, const PbhcConvEncoder* e, float* y, int ldy, int M, void* stream);
size_t pbhc_mlp_packed_floats(int N, int K);
int pbhc_mlp_pack(const float* w, int N, int K, float* packed, void* stream);
size_t pbhc_mlp_fwd_lds_bytes(const int* dims, int num_layers);
int pbhc_mlp_fwd(const float* x, int ldx, const float* const* weights, const float* const* biases, const int* dims, int num_layers, int act, float* y, int ldy,
                 int M, void* stream);
/* The input gradient of that Linear with the activation backward of the layer BELOW folded in (what autograd runs as mm + elu_backward /
 * silu_backward + a column sum; replaces `dy @ w` + pbhc_act_bwd_partials):
 *   dx[M,N] = (dy[M,K] . w[K,N]) * act'(saved[M,N])   N = in_features, K = out_features; saved = the lower layer's activation OUTPUT
 *                                                     (ELU, ReLU) or PRE-activation (SiLU)
 *   scratch[b, :] = column sums of dx over row block b (*num_row_blocks of them, <= PBHC_ACT_MAX_BLOCKS; finish with pbhc_colsum_final)
 * scratch may be NULL (no column sums); act 0: saved unused. */
int pbhc_linear_dgrad_act(const float* dy, const float* w, const float* saved, float* dx, float* scratch, int* num_row_blocks, int M, int N, int K,
                          int act, void* stream);
/* The weight gradient of that Linear (autograd's `dy.t() @ x`): dw[N,K] = sum over the M rows of dy[m,N]^T x[m,K], N = out_features,
 * K = in_features.  The rows are split over pbhc_linear_wgrad_parts(M, N, K) workgroup groups (0: shape not supported — N % 4, M % 32, K < 4 —
 * use the library) whose partial images go to `scratch` (parts * N * K floats) and are summed in a fixed order. */
int pbhc_linear_wgrad_parts(int M, int N, int K);
int pbhc_linear_wgrad(const float* dy, const float* x, float* dw, float* scratch, int M, int N, int K, void* stream);
/* test / measurement aid for the Linear entries above.  shape < 0: everything automatic.  Otherwise bits 0-7: tile shape (0: 128x128, 1: 96x128,
 * 2: 64x128, 3: 64x64 forward only, 0xff: automatic); bits 8-15: K-loop ablation flags (tools/gemm_ablation.py; results are wrong with parts
 * off); bits 16-23: staging version (0: automatic, 1: register-staged version 1, 2: LDS-DMA with BK 16 x 3 stages). */
void pbhc_gemm_debug_force_shape(int shape);

/* The minibatch shuffle of one update (RolloutStorage.mini_batch_generator, agents/modules/data_utils.py:134-152: `tensor.flatten(0, 1)[indices]`
 * per stored key) for several keys in ONE launch: dst_j[i, :] = src_j[index[i], :] for every job j and row i < nrows.  src rows are src_pitch floats
 * apart (the rollout slabs are 128-byte-padded), dst is contiguous [nrows, width]; f32 only; index values in [0, source rows). */
#define PBHC_MAX_GATHER_JOBS 12
typedef struct PbhcGatherJob {
  const float* src;
  float* dst;
  int32_t width;
  int32_t src_pitch;
} PbhcGatherJob;
int pbhc_gather_rows(const PbhcGatherJob* jobs, int num_jobs, const int64_t* index, int nrows, void* stream);

/* nn.utils.clip_grad_norm_(max_norm) + torch.optim.Adam.step() (mh_ppo.py:519-524; weight_decay > 0: torch.optim.AdamW, decoupled,
 * ppo_mimic.py:184-190,682-686) over ONE flat fp32 segment of n
 * parameters (param/grad/exp_avg/exp_avg_sq flat views; lr and step are device scalars, step is incremented).
 * scratch: 512 doubles.  norm_out (may be NULL): the pre-clip gradient norm. */
int pbhc_adam_clip(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int n, const float* lr, float* step, float max_norm, float beta1,
                   float beta2, float eps, float weight_decay, double* scratch, float* norm_out, void* stream);
/* Two consecutive segments (actor [0,n0), critic [n0,n0+n1): MHPPO's two clip_grad_norm_ + two Adam steps, mh_ppo.py:519-524) in ONE launch
 * pair: lr, step, norm_out are 2-element device arrays, scratch 2 x 512 doubles; each segment is clipped by its own norm.
 * zero_grad != 0: the gradient buffer is left ZEROED instead of holding the clipped gradients (the next optimiser step's
 * `optimizer.zero_grad()`, mh_ppo.py:513-514, done by the pass that read the gradients last: no fill launch of its own). */
int pbhc_adam_clip2(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int n0, int n1, const float* lr, float* step, float max_norm, float beta1,
                    float beta2, float eps, float weight_decay, int zero_grad, double* scratch, float* norm_out, void* stream);

/* Test-only: the device functions of csrc/pbhc_math.h (the quaternion / rotation algebra every kernel inlines; SURVEY 8 row a1:
 * isaac_utils/rotations.py:28-669, utils/torch_utils.py:51-79,239-296) applied elementwise, so that they can be pinned directly against
 * the reference's own outputs.  a / b / c: device operand arrays of n elements ([n,4] xyzw quaternions, [n,3] vectors, [n] scalars,
 * [n,9] row-major matrices, as the function takes them); out: [n, width of the result]. */
enum PbhcDebugFn {
  PBHC_DBG_QUAT_ROTATE = 0,        /* a q, b v -> [n,3] */
  PBHC_DBG_QUAT_ROTATE_INVERSE,    /* a q, b v -> [n,3] */
  PBHC_DBG_QUAT_APPLY,             /* a q, b v -> [n,3] */
  PBHC_DBG_QUAT_MUL,               /* a q, b p -> [n,4] */
  PBHC_DBG_QUAT_CONJ,              /* a q -> [n,4] */
  PBHC_DBG_SLERP,                  /* a q0, b q1, c t -> [n,4] */
  PBHC_DBG_CALC_HEADING,           /* a q -> [n] */
  PBHC_DBG_CALC_HEADING_QUAT,      /* a q -> [n,4] */
  PBHC_DBG_CALC_HEADING_QUAT_INV,  /* a q -> [n,4] */
  PBHC_DBG_EULER_XYZ,              /* a q -> [n,3] roll pitch yaw */
  PBHC_DBG_QUAT_FROM_ANGLE_AXIS,   /* a angle [n], b axis -> [n,4] */
  PBHC_DBG_QUAT_ANGLE,             /* a q -> [n]: quat_to_angle_axis(q)[0] */
  PBHC_DBG_AXIS_ANGLE_TO_QUAT_WXYZ,/* a axis-angle [n,3] -> [n,4] wxyz */
  PBHC_DBG_QUAT_TO_MATRIX,         /* a q (xyzw) -> [n,9] */
  PBHC_DBG_MATRIX_TO_QUAT,         /* a [n,9] -> [n,4] xyzw */
  PBHC_DBG_YAW_QUAT,               /* a q -> [n,4] */
  PBHC_DBG_QUAT_TO_MAT6,           /* a q -> [n,6]: first two columns of R, row-major */
  PBHC_DBG_QUAT_UNIT,              /* a q -> [n,4] */
  PBHC_DBG_NUM
};
int pbhc_debug_rotations(int fn, const float* a, const float* b, const float* c, int n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
