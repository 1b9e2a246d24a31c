/* bbai.h -- C ABI of the MI355X batched BabyAI environment engine (libbbai_hip.so).
 *
 * The reference (mila-iqia/babyai) is pure Python and has no FFI; this ABI is the
 * native boundary UNDER the Python adapter that mirrors the reference's env protocol.
 * Each entry point names the reference interface it replaces (file:line under
 * /root/reference).  All device pointers are caller-owned HBM buffers (e.g. the
 * data_ptr() of torch tensors on the handle's device); launches are asynchronous on
 * the caller's hipStream_t (passed as void*; NULL = default stream).  No exceptions
 * cross the boundary: every call returns 0 or a negative bbai_status.  One handle per
 * device; a handle is not thread-safe and follows ONE caller stream at a time: a call that arrives on a different
 * stream than the previous one is ordered (by an event) behind everything the handle enqueued before.  By default that
 * event is recorded on the previous stream at the moment of the switch, so the previous stream must still exist then;
 * see bbai_set_call_events for callers that create and destroy streams.
 */
#ifndef BBAI_H
#define BBAI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum bbai_status {
    BBAI_OK = 0,
    BBAI_ERR_ARG = -1,        /* bad argument / unsupported level configuration */
    BBAI_ERR_HIP = -2,        /* HIP runtime error (see bbai_last_error) */
    BBAI_ERR_STATE = -3,      /* call order violated (e.g. step before seed+reset) */
    BBAI_ERR_NOMEM = -4
};

/* Level description: the constructor arguments of a reference level class
 * (babyai/levels/iclr19_levels.py; LevelGen babyai/levels/levelgen.py:262-291) as data.
 * Layout fields (W..rec_bytes) are derived by bbai_fill_layout. */
typedef struct bbai_level_cfg {
    int32_t kind;                       /* 0 = GoTo family, 1 = LevelGen family, 2 = bonus-level scripts */
    int32_t room_size, num_rows, num_cols, num_dists;
    int32_t redball, connect, check_reach, doors_open, all_unique;     /* GoTo family */
    int32_t instr, target, lock, lock_color_excl, dists_per_room, grey_dists;   /* single-instruction levels */
    int32_t script, sp[4];              /* kind 2: bonus_levels.py gen_mission script id + parameters */
    int32_t locations, unblocking, implicit_unlock;                    /* LevelGen */
    int32_t n_action_kinds, action_kinds[4];    /* 0 goto 1 pickup 2 open 3 putnext, in list order */
    int32_t n_instr_kinds, instr_kinds[3];      /* 0 action 1 and 2 seq, in list order */
    double locked_room_prob;
    int32_t W, H, ES, EH, maxo;
    int32_t off_I, off_app, off_pos, off_cont, off_prog, rec_bytes;
} bbai_level_cfg;

typedef struct bbai_env bbai_env;       /* opaque: N envs of one level on one device */

#define BBAI_OBS_BYTES 147              /* uint8[7][7][3], image[view_x][view_y][channel] */
#define BBAI_PIX_BYTES 9408             /* uint8[56][56][3], RGBImgPartialObsWrapper, tile 8 */
#define BBAI_PROG_BYTES 112             /* compiled instruction tree (mission descriptor) */
#define BBAI_TILE_BYTES 192

int bbai_version(void);
const char* bbai_last_error(void);

/* Derive the record layout for a level configuration. */
int bbai_fill_layout(bbai_level_cfg* cfg);

/* gym.make(id) x n_envs (babyai/levels/levelgen.py:467-493 registration; scripts/train_rl.py:53-60
 * builds the env list).  Allocates all per-env state in HBM on `device`.
 * State layout (an internal choice, never visible in results; bbai_get_option "inplace" reports it, BBAI_INPLACE=1 / 0 forces it): every
 * env owns a ring of pre-generated levels (its RNG stream, generated ahead of need).  Classic: a live record per env, a finished env's
 * next level is copied out of its ring slot.  In-place: the live record IS the ring slot the episode was generated into and a
 * finished env just moves on to the next slot -- no copy and no second launch behind a step; chosen for single-room levels while the
 * ring stays under 12 GiB (the reset-heavy single-room batches: 2 % of the envs finish on every step there), see DESIGN.md section 5. */
int bbai_create(const bbai_level_cfg* cfg, int64_t n_envs, int device, bbai_env** out);
void bbai_destroy(bbai_env* env);

/* env.seed(s) for every env (scripts/train_rl.py:59, babyai/evaluate.py:67-68,105-106):
 * seeds_host[i] (8 bytes per env cross PCIe) -> gym-style sha512 -> MT19937 init_by_array, per lane on the device;
 * then the first levels of every env's stream are generated ahead of need.  Synchronous. */
int bbai_seed(bbai_env* env, const uint64_t* seeds_host, int64_t n);

/* env.reset() for every env (babyai/levels/levelgen.py:35-47; ParallelEnv.reset
 * babyai/rl/utils/penv.py:39-43; ManyEnvs.reset babyai/evaluate.py:68-71): generates the next
 * level of each env's RNG stream on the device and writes the first observation. */
int bbai_reset(bbai_env* env, uint8_t* image_dev, uint8_t* dir_dev, void* stream);

/* env.step(action) for every env (babyai/levels/levelgen.py:49-66).
 *   auto_reset != 0 : ParallelEnv semantics (babyai/rl/utils/penv.py:8-11,45-52): a finished env
 *                     is reset in the same call and image/dir hold the NEXT episode's first obs,
 *                     while reward/done are the terminal step's.
 *   auto_reset == 0 : ManyEnvs semantics (babyai/evaluate.py:73-81): a finished env is frozen and
 *                     keeps re-emitting its last (obs, reward, done) until bbai_reset.
 * actions_dev[i] is 0..6 (MiniGridEnv.Actions), or BBAI_ACTION_RESET_ENV = "env.reset() for this env now": the
 * episode is abandoned with done = 1, reward = 0 and handled like any finished env above (a ParallelEnv worker's
 * `reset` command, penv.py:12-14; scripts/make_agent_demos.py:84-88 after a bot crash).
 * UNKNOWN ACTIONS: the reference asserts on an action outside its enum (gym_minigrid MiniGridEnv.step: `assert False,
 * "unknown action"`, reached from levelgen.py:50).  A kernel cannot raise: bytes 8..255 are DEFINED here as the `done`
 * action (no movement; the step is counted, the verifier sees a done action).  Callers that want the reference's
 * behaviour check their actions first: the Python binding does (BatchedBabyAIEnv(validate_actions=True) / step(...,
 * validate=True): one device-side max-reduce and a host sync, off by default; the list-of-dicts adapters, whose actions
 * are host arrays anyway, always check and raise AssertionError("unknown action")).
 * reward_dev[i]   = float32 rounding of the reward (what babyai/rl/algos/base.py:162-167 makes of it);
 * reward64_dev[i] = the reference's own return value: MiniGridEnv._reward() as a Python float (levelgen.py:59-61),
 *                   bit for bit (what babyai/evaluate.py:128 accumulates).  May be NULL. */
#define BBAI_ACTION_RESET_ENV 7
int bbai_step(bbai_env* env, const uint8_t* actions_dev, uint8_t* image_dev, uint8_t* dir_dev,
              float* reward_dev, double* reward64_dev, uint8_t* done_dev, int auto_reset, void* stream);

/* RGBImgPartialObsWrapper.observation (gym_minigrid.wrappers; used at babyai/evaluate.py:91-92,
 * scripts/train_rl.py:57-58): encoded obs uint8[N][147] -> pixels uint8[N][56][56][3].
 * The tile atlas must have been installed with bbai_set_atlas. */
int bbai_set_atlas(bbai_env* env, const uint8_t* tiles_host, int n_tiles, const uint8_t* lut_host /* [2][256] */);
int bbai_render(bbai_env* env, const uint8_t* image_dev, uint8_t* pixels_dev, void* stream);

/* env.step(action) of a batch wrapped in RGBImgPartialObsWrapper (babyai/evaluate.py:91-92: `env = RGBImgPartialObsWrapper(env)`, whose
 * step() returns the rendered observation) as ONE call: exactly bbai_step followed by bbai_render(image_dev -> pixels_dev), same bytes.
 * With option "step_render_split" = 1 (and a batch of 262 144 envs or more, auto-reset consumed inside the step kernel) the batch is stepped
 * in two halves and the second half's step kernel runs on a stream of the handle's own UNDER the first half's render; everything is joined
 * on `stream` before the call's last launch, so callers see no difference but the time. */
int bbai_step_render(bbai_env* env, const uint8_t* actions_dev, uint8_t* image_dev, uint8_t* dir_dev, float* reward_dev,
                     double* reward64_dev, uint8_t* done_dev, int auto_reset, uint8_t* pixels_dev, void* stream);

/* Mission text as token ids, device-resident (replaces the per-step regex tokenisation of every mission in
 * InstructionsPreprocessor, babyai/utils/format.py:59-75): register a caller-owned uint8[N][72] buffer; the engine
 * rewrites env i's row whenever env i starts a new episode.  Ids follow babyai_amd/missions.py VOCAB, 0 = padding. */
#define BBAI_TOK_MAX 72
int bbai_set_token_buffer(bbai_env* env, uint8_t* tokens_dev);

/* State access (host buffers; synchronous): parity tests, checkpoints, mission strings.  hot_host[15] is the env's place in its
 * look-ahead ring: engine bookkeeping of the EXPORTING handle -- bbai_import_state ignores it and keeps the importing handle's own
 * (the ring belongs to the handle; in the in-place layout the slot says where the live record is). */
int bbai_export_state(bbai_env* env, int64_t first, int64_t count, uint8_t* rec_host,
                      uint8_t* hot_host /* 16 B each */, uint64_t* stale_host);
int bbai_import_state(bbai_env* env, int64_t first, int64_t count, const uint8_t* rec_host,
                      const uint8_t* hot_host, const uint64_t* stale_host);
int bbai_get_programs(bbai_env* env, int64_t first, int64_t count, uint8_t* prog_host /* 112 B each */);

/* Checkpoint / resume of a whole batch (the reference checkpoints only its model, babyai/utils/model.py:29-32; an
 * auto-resetting env batch additionally needs its RNG streams): the blob holds the live state, every env's MT19937
 * stream, the look-ahead ring with its window bookkeeping, the counters and -- when bbai_bot_act has been used -- the
 * expert's plans.  Loading it into a fresh handle of the same level and batch size continues the run
 * bit-identically, auto-resets included (tests/test_gpu_parity.py::test_checkpoint_resume_*); a handle whose look-ahead
 * period differs (it is chosen from the free memory at bbai_create unless BBAI_LOOKAHEAD pins it) takes the blob's ring shape; a blob of the
 * other state layout (bbai_create), of the other done-action mode, of another expert stack capacity, of another format version or of the
 * wrong size is refused BEFORE the handle is touched: a refused load leaves the handle exactly as it was.  Synchronous, host buffers.
 * Caller-owned buffers are not part of the blob: keep the last observation next to it if it is needed before the next
 * step, and register the token buffer again after a load (bbai_set_token_buffer refills every row of a live handle). */
int64_t bbai_checkpoint_bytes(bbai_env* env);
int bbai_checkpoint_save(bbai_env* env, void* host_buf, int64_t bytes);
int bbai_checkpoint_load(bbai_env* env, const void* host_buf, int64_t bytes);

/* The reference's GOFAI expert for every env: one `Bot.replan(action_taken)` decision each
 * (babyai/bot.py:547-597; callers babyai/utils/agent.py:139-146 BotAgent.act, scripts/make_agent_demos.py:93-107).
 * `prev_actions_dev` = the action each env was actually stepped with since the previous call (advising mode,
 * bot.py:88-98), or NULL = "the suggestion was taken" (replan(None)).  An env whose episode has just started
 * (step_count == 0), or whose expert was not consulted on the previous step, gets a fresh Bot (= `Bot(env)` there).  `actions_dev[i]` = suggested action 0..6, or 255 where the reference bot would
 * have raised (assertion / DisappearedBoxError / endless replanning); it stays 255 until the episode ends.
 * Decision-for-decision parity with the reference bot: tests/test_hostsim_bot.py, tests/golden/bot/.
 * The expert's plan lives in the handle (allocated on first use, ~1.7 KB per env) and is not part of
 * bbai_export_state / bbai_import_state: import states at episode boundaries when the expert is in use. */
int bbai_bot_act(bbai_env* env, const uint8_t* prev_actions_dev, uint8_t* actions_dev, void* stream);
/* The inner loop of expert-driven demonstration generation (scripts/make_agent_demos.py:71-137 generate_demos with
 * BotAgent, babyai/utils/agent.py:139-155) for every env, T steps per call with NO host round trip between the steps:
 *   for t in 0..T-1:  images_out[t], dirs_out[t] (, tokens_out[t]) = the observation (and mission tokens) the expert decides on;
 *                     actions_out[t] = Bot.replan(None) -- a bot that gave up emits BBAI_ACTION_RESET_ENV ("env.reset() on the
 *                     same stream", make_agent_demos.py:84-88) and gave_up_out[t] = 1 there;
 *                     step with ParallelEnv semantics (auto-reset); rewards_out[t], dones_out[t].
 * image_dev / dir_dev hold the CURRENT observation on entry (what the last reset / step / rollout wrote) and on return.
 * History buffers are [T][n_envs] rows of the per-step layouts (147 / 1 / 72 / 1 / 4 / 1 / 1 bytes per env).  tokens_out may
 * be NULL; otherwise a token buffer must be registered (bbai_set_token_buffer).  The expert and the step are the kernels of
 * bbai_bot_act / bbai_step: same decisions, same bytes (tests/test_gpu_parity.py::test_bot_rollout_*). */
int bbai_bot_rollout(bbai_env* env, int T, uint8_t* image_dev, uint8_t* dir_dev, uint8_t* images_out, uint8_t* dirs_out,
                     uint8_t* tokens_out, uint8_t* actions_out, float* rewards_out, uint8_t* dones_out, uint8_t* gave_up_out,
                     void* stream);
/* Bots that gave up so far: by the reference's own rules / because a fixed-size structure of this port overflowed
 * (subgoal stack, 48 entries unless the environment variable BBAI_BOT_STACK says otherwise when the expert is first
 * used; same-colour keys 12).  The reference's stack is an unbounded list: a bot that replans for ever inside one
 * decision, or loops without progress across steps until max_steps, overflows here instead (observed: UnlockToUnlock,
 * 1 in 1024 MiniBossLevel missions); those episodes fail in the reference too. */
int bbai_bot_stats(bbai_env* env, uint64_t* gave_up, uint64_t* capacity);

/* Measurement aid: copy the outputs of the first `count` envs of a batch (image 147 B, direction, float64 reward, done) and
 * the pixel images of the first `pix_count` into caller-owned log rows with ONE launch on `stream` -- the in-run parity tap of
 * bench.py ("first 1024 envs of every shard, every step, all outputs").  pixels / pixels_out may be NULL when
 * pix_count is 0; both must be 16-byte aligned otherwise.  bbai_tap and bbai_gae take no handle: like any HIP call
 * without one they launch on the calling thread's CURRENT device, which must own `stream` and every pointer. */
int bbai_tap(int64_t count, int64_t pix_count, const uint8_t* image_dev, const uint8_t* dir_dev, const double* reward64_dev,
             const uint8_t* done_dev, const uint8_t* pixels_dev, uint8_t* image_out, uint8_t* dir_out, double* reward64_out,
             uint8_t* done_out, uint8_t* pixels_out, void* stream);
/* The same for an arbitrary list of envs: ids_dev int64[count] on the device, env ids_dev[k] -> log row k (pixels of the
 * first pix_count <= count listed envs).  bench.py taps ids scattered over the whole shard (both ends, block and wave
 * boundaries, a pseudo-random spread: babyai_amd/shard.py scattered_ids). */
int bbai_tap_ids(int64_t count, int64_t pix_count, const int64_t* ids_dev, const uint8_t* image_dev, const uint8_t* dir_dev,
                 const double* reward64_dev, const uint8_t* done_dev, const uint8_t* pixels_dev, uint8_t* image_out, uint8_t* dir_out,
                 double* reward64_out, uint8_t* done_out, uint8_t* pixels_out, void* stream);

/* A step that logs its own tap rows.  bbai_step_tap_set lists the envs (host array, any order, no duplicates; count 0 clears); then
 * bbai_step_tapped = bbai_step + "bbai_tap_ids of the listed envs into these rows" (image_out uint8[count][147], dir_out uint8[count],
 * reward64_out double[count], done_out uint8[count]; log row k = env ids[k]) -- with the rows written by the stepping lanes themselves
 * wherever the step kernel leaves the final outputs behind (fused consume, in-place layout, auto_reset 0): no launch behind the step.  A
 * tap launch is 3 us + a dependent-launch gap, a quarter of a 65 536-env step; a logger that follows a few envs through every step
 * (bench.py's in-run parity check, a monitor) should not cost that.  Unfused auto-resetting steps (option consume_fused 0) get the
 * launch behind k_consume: same bytes either way (tests/test_gpu_parity.py::test_step_tapped_equals_step_plus_tap).  reward64 must not
 * be NULL here.  BBAI_ERR_STATE before bbai_step_tap_set. */
int bbai_step_tap_set(bbai_env* env, const int64_t* ids_host, int64_t count);
int bbai_step_tapped(bbai_env* env, const uint8_t* actions_dev, uint8_t* image_dev, uint8_t* dir_dev, float* reward_dev, double* reward64_dev,
                     uint8_t* done_dev, int auto_reset, uint8_t* image_out, uint8_t* dir_out, double* reward64_out, uint8_t* done_out, void* stream);

/* An open-loop rollout: T steps with NO host round trip in between.  actions_dev[T][n_envs] are resident on the device -- the
 * random-action rollouts the reference's own level test plays (babyai/levels/levelgen.py:522-527), a replayed demonstration, a
 * benchmark's pre-drawn action stream.  For t = 0 .. T-1 exactly what a caller's loop would enqueue:
 *     bbai_step(actions_dev + t * n_envs, ...)                       (auto_reset as in bbai_step)
 *     bbai_render(image_dev -> pixels_dev)                           when pixels_dev != NULL
 *     bbai_tap_ids(...) into log rows obs_row0 + t / row0 + t        when tap != NULL
 * image / dir / reward / reward64 / done (/ pixels) hold the LAST step's outputs on return; the tap log keeps every step of the listed
 * envs: image_out [rows][count][147], dir_out [rows][count], pixels_out [rows][pix_count][9408] indexed by obs_row0 + t (a caller
 * that stored the reset()'s observation in row 0 passes obs_row0 = 1), reward64_out / done_out [rows][count] indexed by row0 + t.
 * The per-step calls give the same bytes (tests/test_gpu_parity.py::test_rollout_entry_*).
 * Since round 6 the entry is also the fast path of open-loop stepping: with encoded observations (pixels_dev NULL), no token buffer,
 * and steps that move their finished envs on themselves (the default: fused consume / in-place layout; or auto_reset 0), ONE k_step
 * launch takes every step the current look-ahead window has left (up to 64 single-room / 96 maze ticks) -- an env's step touches only its own state, so a 64-env
 * block walks through the ticks on its own, no launch boundary and no dependent-launch gap (4-5 us: a third of a 65 536-env step) in
 * between; tick t reads actions_dev + t * n_envs.  A tap log is then written by the stepping lanes (bbai_step_tapped's mechanism): pass
 * ids_dev = NULL and count = the number of envs listed by bbai_step_tap_set (log row k = env ids[k]; pix_count 0).  A tap log WITH
 * ids_dev is a bbai_tap_ids launch behind every step and keeps one step per launch.  Option "rollout_multi" 0 (BBAI_ROLLOUT_MULTI=0):
 * one step per launch everywhere. */
typedef struct bbai_tap_log {
    int64_t count, pix_count;           /* envs listed / how many of the first listed ones also log pixels */
    const int64_t* ids_dev;             /* int64[count] env indices; NULL: the envs of bbai_step_tap_set (count must match) */
    uint8_t* image_out; uint8_t* dir_out; double* reward64_out; uint8_t* done_out; uint8_t* pixels_out;
    int64_t obs_row0, row0;
} bbai_tap_log;
int bbai_rollout(bbai_env* env, int T, const uint8_t* actions_dev, uint8_t* image_dev, uint8_t* dir_dev, float* reward_dev,
                 double* reward64_dev, uint8_t* done_dev, int auto_reset, uint8_t* pixels_dev, const bbai_tap_log* tap, void* stream);

/* Generalised advantage estimation of a rollout on the current device (the loop of babyai/rl/algos/base.py:196-202 as
 * one reverse scan per env).  All buffers float32, env-major [num_envs][num_frames] (the order base.py:207-232 flattens
 * experiences to); masks[p][i] = 1 - done before frame i, last_mask / last_value [num_envs] = the mask and the critic's
 * value after the last frame.  Writes advantage and returnn = value + advantage.  float32 in the reference's operation
 * order: bit-identical to its torch loop. */
int bbai_gae(int64_t num_envs, int num_frames, const float* rewards_dev, const float* values_dev, const float* masks_dev,
             const float* last_mask_dev, const float* last_value_dev, double discount, double gae_lambda,
             float* advantage_dev, float* returnn_dev, void* stream);

/* Per-kernel timing for measurements (bench.py's roofline): while enabled, every k_step / k_consume / k_render launch is
 * bracketed by a HIP event pair ON THE STREAM IT IS LAUNCHED ON; bbai_profile_read returns the summed milliseconds and the
 * launch counts in that order.  enable: 1 = start from zero, 2 = resume (totals kept), 0 = pause (totals stay readable).
 * Costs two event records per launch.  A bbai_rollout launch of k_step takes several steps: bbai_get_option "profile_step_ticks" =
 * the steps the bracketed k_step launches took. */
int bbai_profile(bbai_env* env, int enable);
int bbai_profile_read(bbai_env* env, double* ms_total /* [3] */, int64_t* launches /* [3] */);

/* Stream switching policy.  enable != 0: every reset / step / render / bot_act ends by recording the handle's completion
 * event on its own stream, and a call on another stream only waits for that event -- the previous stream is never
 * touched again and may have been destroyed.  Costs one event record per call (measured +3.4 us per step at 65 536 envs,
 * profiles/r03/call_events_ab.jsonl), hence off by default (BBAI_CALL_EVENTS=1 in the environment turns it on at create). */
int bbai_set_call_events(bbai_env* env, int enable);

/* The reference's done-action verifier mode (babyai/levels/verifier.py:17 `use_done_actions = os.environ.get(
 * 'BABYAI_DONE_ACTIONS', False)`, :216-230 ActionInstr.verify): an action instruction succeeds only on a `done` action taken
 * right after the step that completed it, and a `done` action at any other time FAILS it (episode over, reward 0).  A handle
 * starts in that mode iff the variable is non-empty in the environment at bbai_create -- the reference reads it at import --
 * and this call switches it explicitly (every env's lastStepMatch is cleared; call it between episodes).
 * AndInstr's extra failure rule (verifier.py:543-545: both of its instructions fail on a `done` => the And fails) sits behind
 * `action is self.env.actions.done`, an IDENTITY test: an int 6 never passes it (every vectorised caller of the reference steps
 * with ints, babyai/rl/utils/penv.py:8), the enum member does -- and the reference's own expert returns the member (babyai/bot.py:593,
 * fed to env.step by scripts/make_agent_demos.py:93-107).  A byte cannot carry that difference, so the caller says which it is:
 * bbai_bot_rollout applies the rule (its actions are the expert's), bbai_step applies it iff bbai_set_option(env,
 * "done_action_enum", 1) -- for a loop that steps with what bbai_bot_act suggested.  Pinned by traces of the reference stepped with
 * the enum member (tests/golden/done_actions_enum/, tests/test_done_actions.py).
 * The per-env bits travel in checkpoints; bbai_import_state clears them. */
int bbai_set_done_actions(bbai_env* env, int enable);
int bbai_get_done_actions(bbai_env* env);

/* Performance knobs of a live handle, by name (what the BBAI_* environment variables set at bbai_create; the reference has
 * no counterpart: these choose launch shapes and buffers, never results -- every setting but the last yields the same bytes, which
 * tests/test_gpu_parity.py::test_options_do_not_change_results checks).  Synchronises the device.  Names:
 *   "render_queue"      -1 = by batch size (default), 0 = one-shot render blocks, m > 0 = persistent-block queue shape m
 *   "render_pace"       experiment: 1/16 ns of wall clock per render ticket (a time gate over two ticket counters); 0 = off (default)
 *   "render_queue_bpc", "render_queue_blocks"   persistent render blocks per CU (0 = 1024 threads' worth) / in total (0 = per CU)
 *   "render_group", "render_tpb"   envs / threads per one-shot render block (0 = by batch size)
 *   "step_prio", "pregen_group", "pregen_blocks", "pregen_min", "consume_fused"   as BBAI_STEP_PRIO / BBAI_PREGEN_GROUP /
 *                       BBAI_PREGEN_BLOCKS / BBAI_PREGEN_MIN / BBAI_CONSUME_FUSED
 *   "step_render_split" bbai_step_render / bbai_rollout with pixels: 1 = step the batch in two halves, the second under the first half's
 *                       render; 0 = never; -1 = the library's default
 *   "gate_strict"       1 = the step stream ALSO waits, at the start of every look-ahead window, for the refill launched two windows
 *                       earlier (rounds 1-4's rule); 0 (default) = it runs ahead of the refills as far as every env is sure to keep
 *                       a window's worth of ready levels (k_gate) -- a reset storm then refills under the steps
 *   "gate_probe"        1 (default; BBAI_GATE_PROBE) = the first window a caller's stream opens probes whether its kernels run concurrently with the
 *                       look-ahead stream's (HIP does not promise it: streams may share a hardware queue, a profiler may serialise
 *                       launches); a stream that fails runs under the strict rule.  0 = no probe, 2 = every probe fails (tests).  Setting
 *                       it forgets the verdicts so far
 *   "pregen_per_group"  single-room levels: list entries per working lane group of a refill launch (BBAI_PREGEN_PER_GROUP, default 12)
 *   "pregen_lane"       which kernel generates the look-ahead levels (results identical): 1 = k_pregen_lane, one lane per level (bit boards and an
 *                       object table instead of planes; every LevelGen parameterisation and the single-instruction levels without a lock-first
 *                       prologue; default for the single rooms), 0 = k_pregen, one lane group per level (every kind; default elsewhere);
 *                       BBAI_PREGEN_LANE at bbai_create.  Setting it converts the handle's MT19937 states between the two kernels' forms.
 *                       BBAI_ERR_ARG when 1 is asked of a kind the lane generator does not cover
 *   "lane_blocks"       upper bound on k_pregen_lane's waves per launch (BBAI_LANE_BLOCKS, default 16 384)
 *   "lookahead_streams" look-ahead streams a window's refill is split over (1 .. 8, default 1; BBAI_LOOKAHEAD_STREAMS): each takes a contiguous range of
 *                       64-env blocks, so an env's levels stay in stream order on ONE stream while the ranges' refills run side by side.  Measured: no gain
 *                       (a refill launch lasts as long as its slowest wave whatever its size) -- and the extra streams are created only while asked
 *                       for: every HIP stream is a hardware queue, a dozen per handle slow EVERY dispatch down
 *   "rollout_multi"     bbai_rollout with encoded observations: 1 (default; BBAI_ROLLOUT_MULTI) = one k_step launch per look-ahead window's remaining steps,
 *                       0 = one launch per step
 *   "gate_fault_inject" tests: raise (1) / clear (0) the sticky word a timed-out window gate leaves behind
 *   "bot_group"         the expert's kernel (bbai_bot_act / bbai_bot_rollout): 0 (default) = one lane per env (k_bot), 16 = one 16-lane
 *                       group per env with the first search in LDS (k_botg: same decisions, measured ~2 x slower -- an experiment
 *                       kept for reference); also BBAI_BOT_GROUP at bbai_create
 *   "done_action_enum"  the one semantic switch, meaningful in done-action mode only: see bbai_set_done_actions
 * BBAI_ERR_ARG for an unknown name. */
int bbai_set_option(bbai_env* env, const char* name, int64_t value);
/* Read a knob back (the names of bbai_set_option), or "lookahead_period" (the refill period the handle chose at bbai_create), or
 * "inplace" (1: the in-place state layout, bbai_create), or "gate_timeouts" (synchronises: window gates that gave up waiting for a
 * look-ahead refill after ~10 s -- must be 0; anything else means a refill was lost and the batch's results are void), or
 * "gate_fault" (the same fact as a sticky flag in pinned host memory, read without synchronising: once it is set EVERY stepping /
 * resetting entry point of the handle returns BBAI_ERR_STATE before it enqueues anything, until bbai_seed regenerates the ring),
 * or "gate_forced_strict" (1: the caller's stream failed the concurrency probe and runs under the strict rule). */
int bbai_get_option(bbai_env* env, const char* name, int64_t* out);

/* Number of level generations (resets) performed so far, all envs. */
int bbai_reset_count(bbai_env* env, uint64_t* out);

/* Levels the generator gave up on (last-resort guard after 200 000 rejected attempts; the env is frozen).  Expected 0:
 * every unbounded rejection loop of the reference that can spin for ever is bounded explicitly. */
int bbai_generator_failures(bbai_env* env, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
