/* orc_prefix.h — TEST INFRASTRUCTURE. Renames every entry point of include/rnb_neus2.h to its orc_ twin so the CPU oracle
 * (liborc.so) can export the same signatures beside the HIP library, and so tests/ can build the host programs
 * (e.g. the testbed CLI) against the oracle with `g++ -include oracle/orc_prefix.h`. Never included by the product. */
#ifndef ORC_PREFIX_H
#define ORC_PREFIX_H
#define rnb_last_error orc_last_error
#define rnb_abi_version orc_abi_version
#define rnb_default_config orc_default_config
#define rnb_create orc_create
#define rnb_destroy orc_destroy
#define rnb_update_config orc_update_config
#define rnb_n_params orc_n_params
#define rnb_param_layout orc_param_layout
#define rnb_grid_tables orc_grid_tables
#define rnb_init_params orc_init_params
#define rnb_set_params orc_set_params
#define rnb_buffer orc_buffer
#define rnb_params_changed orc_params_changed
#define rnb_bitfield_changed orc_bitfield_changed
#define rnb_sdf_lattice orc_sdf_lattice
#define rnb_marching_cubes orc_marching_cubes
#define rnb_memcpy orc_memcpy
#define rnb_device_malloc orc_device_malloc
#define rnb_device_free orc_device_free
#define rnb_set_dataset orc_set_dataset
#define rnb_set_training_step orc_set_training_step
#define rnb_valid_level orc_valid_level
#define rnb_update_density_grid orc_update_density_grid
#define rnb_update_density_bitfield orc_update_density_bitfield
#define rnb_update_density_grid_begin orc_update_density_grid_begin
#define rnb_update_density_grid_end orc_update_density_grid_end
#define rnb_set_grid_exchange orc_set_grid_exchange
#define rnb_density orc_density
#define rnb_sdf orc_sdf
#define rnb_forward_infer orc_forward_infer
#define rnb_generate_training_samples orc_generate_training_samples
#define rnb_compute_loss orc_compute_loss
#define rnb_forward_backward orc_forward_backward
#define rnb_optimizer_step orc_optimizer_step
#define rnb_train_step orc_train_step
#define rnb_train_step_begin orc_train_step_begin
#define rnb_train_step_end orc_train_step_end
#define rnb_train_step_apply orc_train_step_apply
#define rnb_train_step_local orc_train_step_local
#define rnb_train_step_finish orc_train_step_finish
#define rnb_training_step orc_training_step
#define rnb_profile_enable orc_profile_enable
#define rnb_profile_count orc_profile_count
#define rnb_profile_get orc_profile_get
#define rnb_rays_per_batch orc_rays_per_batch
#define rnb_set_controller orc_set_controller
#define rnb_set_optimizer_step orc_set_optimizer_step
#define rnb_eval_primitives orc_eval_primitives
#define rnb_gradient_parts orc_gradient_parts
#define rnb_gradient_part_wait orc_gradient_part_wait
#define rnb_train_step_apply_early orc_train_step_apply_early
#define rnb_shard_layout orc_shard_layout
#define rnb_train_step_apply_shard orc_train_step_apply_shard
#define rnb_train_step_apply_done orc_train_step_apply_done
#define rnb_ctx orc_ctx_s
#endif
