// TEMPORARY stub — replaced by the real implementation
#include "xq_internal.h"
using namespace xq;
extern "C" {
int xq_dqn_create(const int* layer_sizes, int n_sizes, double learning_rate, double gamma, uint64_t seed, void* hip_stream, xq_dqn** out) { return fail(XQ_ERR_RUNTIME, "xq_dqn_create: not implemented yet"); }
int xq_dqn_destroy(xq_dqn* d) { return fail(XQ_ERR_RUNTIME, "xq_dqn_destroy: not implemented yet"); }
int xq_dqn_num_params(const xq_dqn* d, size_t* n_weights, size_t* n_biases) { return fail(XQ_ERR_RUNTIME, "xq_dqn_num_params: not implemented yet"); }
int xq_dqn_set_params(xq_dqn* d, int which_net, const double* weights_host, const double* biases_host) { return fail(XQ_ERR_RUNTIME, "xq_dqn_set_params: not implemented yet"); }
int xq_dqn_get_params(xq_dqn* d, int which_net, double* weights_host, double* biases_host) { return fail(XQ_ERR_RUNTIME, "xq_dqn_get_params: not implemented yet"); }
int xq_dqn_forward(xq_dqn* d, int which_net, const double* states_host, int n, double* q_host) { return fail(XQ_ERR_RUNTIME, "xq_dqn_forward: not implemented yet"); }
int xq_dqn_forward_boards_dev(xq_dqn* d, int which_net, const uint32_t* boards_dev, int n, int n_out, float* q_dev, int ldq) { return fail(XQ_ERR_RUNTIME, "xq_dqn_forward_boards_dev: not implemented yet"); }
int xq_dqn_backpropagate(xq_dqn* d, const double* states_host, const double* targets_host, int n, double learning_rate, double grad_scale, int mode) { return fail(XQ_ERR_RUNTIME, "xq_dqn_backpropagate: not implemented yet"); }
int xq_dqn_update_target(xq_dqn* d) { return fail(XQ_ERR_RUNTIME, "xq_dqn_update_target: not implemented yet"); }
int xq_dqn_save_model(xq_dqn* d, const char* path) { return fail(XQ_ERR_RUNTIME, "xq_dqn_save_model: not implemented yet"); }
int xq_dqn_load_model(xq_dqn* d, const char* path) { return fail(XQ_ERR_RUNTIME, "xq_dqn_load_model: not implemented yet"); }
int xq_dqn_td_grads(xq_dqn* d, const uint32_t* boards_dev, const uint32_t* next_boards_dev, const int32_t* action_to_dev, const float* reward_dev, const uint8_t* done_dev, const int32_t* slots_dev, int n, int td_net, int mode) { return fail(XQ_ERR_RUNTIME, "xq_dqn_td_grads: not implemented yet"); }
int xq_dqn_apply_grads(xq_dqn* d, double learning_rate, double grad_scale) { return fail(XQ_ERR_RUNTIME, "xq_dqn_apply_grads: not implemented yet"); }
int xq_dqn_grad_buffer(xq_dqn* d, float** grads_dev, size_t* n_floats) { return fail(XQ_ERR_RUNTIME, "xq_dqn_grad_buffer: not implemented yet"); }
int xq_dqn_td_grads_replay(xq_dqn* d, xq_replay* r, int batch, int td_net, int mode) { return fail(XQ_ERR_RUNTIME, "xq_dqn_td_grads_replay: not implemented yet"); }
int xq_dqn_td_update_host(xq_dqn* d, int n, const uint8_t* boards90, const uint8_t* next_boards90, const int32_t* action_to, const float* reward, const uint8_t* done, int td_net, int mode, double learning_rate, double grad_scale, float* q_sa_out, float* y_out) { return fail(XQ_ERR_RUNTIME, "xq_dqn_td_update_host: not implemented yet"); }
int xq_dqn_last_loss(xq_dqn* d, double* loss) { return fail(XQ_ERR_RUNTIME, "xq_dqn_last_loss: not implemented yet"); }
int xq_dqn_kernel_stats(xq_dqn* d, int enable, xq_kernel_stat* stats, int max_stats, int* n_stats) { return fail(XQ_ERR_RUNTIME, "xq_dqn_kernel_stats: not implemented yet"); }
}
