// TEMPORARY stub — replaced by the real implementation
#include "xq_internal.h"
using namespace xq;
extern "C" {
int xq_trainer_create(const xq_trainer_config* cfg, void* hip_stream, xq_trainer** out) { return fail(XQ_ERR_RUNTIME, "xq_trainer_create: not implemented yet"); }
int xq_trainer_destroy(xq_trainer* t) { return fail(XQ_ERR_RUNTIME, "xq_trainer_destroy: not implemented yet"); }
int xq_trainer_env(xq_trainer* t, xq_env** env) { return fail(XQ_ERR_RUNTIME, "xq_trainer_env: not implemented yet"); }
int xq_trainer_dqn(xq_trainer* t, xq_dqn** dqn) { return fail(XQ_ERR_RUNTIME, "xq_trainer_dqn: not implemented yet"); }
int xq_trainer_replay(xq_trainer* t, xq_replay** replay) { return fail(XQ_ERR_RUNTIME, "xq_trainer_replay: not implemented yet"); }
int xq_trainer_collect(xq_trainer* t) { return fail(XQ_ERR_RUNTIME, "xq_trainer_collect: not implemented yet"); }
int xq_trainer_learn_grads(xq_trainer* t) { return fail(XQ_ERR_RUNTIME, "xq_trainer_learn_grads: not implemented yet"); }
int xq_trainer_learn_apply(xq_trainer* t, int world_size) { return fail(XQ_ERR_RUNTIME, "xq_trainer_learn_apply: not implemented yet"); }
int xq_trainer_step(xq_trainer* t, int n_iterations) { return fail(XQ_ERR_RUNTIME, "xq_trainer_step: not implemented yet"); }
int xq_trainer_counters(xq_trainer* t, uint64_t* env_steps, uint64_t* updates, uint64_t* episodes) { return fail(XQ_ERR_RUNTIME, "xq_trainer_counters: not implemented yet"); }
}
