"""The CPU checker's floating-point primitives on the hot path (hash-grid index / fraction, NerfCoordinate warps, the activations of the NeuS alpha and of the
albedo, the L1 / L2 ray loss, image / pixel choice of a ray) against the outputs of the REFERENCE's own host-compilable fragments
(tests/golden/float_fixtures.json, written by tests/golden/make_float_fixtures.py in the build container): bit for bit. The same items run through the HIP
library in tests/test_gpu_parity.py."""
from tests import float_fixture_cases, oracle_lib

COUNTS = {"activation": 256, "warp": 192, "loss": 128, "pixel": 256, "grid": 512, "read_rgba": 256, "camera_ray": 192, "ray_targets": 192, "loss_sample": 384, "ray_loss": 192, "encode": 160, "march_ray": 192, "sdf_density": 512, "prep_due": 709}


def test_oracle_float_primitives_match_the_reference_fragments():
    c = oracle_lib.context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10)
    try:
        n = float_fixture_cases.check(c, exact_exp=True)
    finally:
        c.close()
    assert n == COUNTS


def test_oracle_level_tables_match_the_reference_constructor_loop():
    assert float_fixture_cases.check_level_tables(lambda **cfg: oracle_lib.context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, **cfg)) == 10


def test_oracle_valid_level_schedule_matches_the_reference():
    assert float_fixture_cases.check_valid_levels(lambda **cfg: oracle_lib.context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, **cfg)) == 4 * 328


def test_oracle_optimizer_matches_the_reference_kernels():
    c = oracle_lib.context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, n_levels=2)
    try:
        c.init_params()
        assert float_fixture_cases.check_optimizer(c, exact_pow=True) == 256
    finally:
        c.close()


def test_oracle_occupancy_update_samples_match_the_reference_kernel():
    from rnb_neus2_amd import synthetic
    c = oracle_lib.context(target_batch_size=1 << 10, max_rays_per_batch=1 << 10, n_levels=2)
    try:
        c.init_params()
        c.set_dataset(*synthetic.make_scene(2, 16, 28.0))
        assert float_fixture_cases.check_grid_samples(c) == 3 * 512
        assert float_fixture_cases.check_bitfield(c) == 16
    finally:
        c.close()


def test_oracle_ray_batch_controller_matches_the_reference_statements():
    assert float_fixture_cases.check_controller(lambda **cfg: oracle_lib.context(**cfg)) == 256


def test_oracle_learning_rate_schedule_matches_the_reference():
    assert float_fixture_cases.check_lr_decay(lambda **cfg: oracle_lib.context(**cfg)) == 62


def test_float_fixture_is_what_its_generator_says():
    fx = float_fixture_cases.load()
    assert "make_float_fixtures.py" in fx["_source"] and "-ffp-contract=off" in fx["_source"]
    assert set(fx) == {"_source", "activation_val_relu_logistic_rgb_rgbderivative", "warp_lo_hi_p3_d3_dt_warpedp3_unwarpedp3_warpedd3_unwarpedd3_warpeddt_unwarpeddt",
                       "loss_isL2_target4_prediction4_loss_gradient4", "pixel_base_nrays_total_nimg_w_h_snap_advlo_advhi_img_x_y", "grid_size_res_pg3_index0_index1_x_scale_pos_cell",
                       "levels_n_base_log2hash_scalebits_offsets_resolutions_scales", "validlevel_n_basescale_scale_basestep_step_level",
                       "readrgba_w_h_x_y_pixels28_rgba4_rednonpositive", "axes_mode_scale_offset3_matrix12_ngp12", "cameraray_w_h_focal2_pp2_xy2_xform12_o3_d3_dir3",
                       "raytargets_flags5_light_xform12_texnormal4_texalbedo4_lightdirs9_rgbtarget4_light3_normal3_shading_supernormal",
                       "losssample_flags4_out8_in25_alpha_T_w2_rgb4_dl11_shading_ek_inter10",
                       "rayloss_L2_rgbplus_bce_maskweight_nrays_target4_ray4_albedoalpha_normalalpha_weightsum_loss_grad4_ws_gws_lossrow_maskrow",
                       "adam_globals8_then_ismatrix_step_optstep_w_w16_g16_m_v_ema16_neww_neww16_newm_newv_newstep_newema16",
                       "encode_size_res_scale_xyz_table257_f0_f1_dydx6", "gridsamples_call_slot_idx_pos3", "bitfield_pattern_mean_table8_then_setbits_checksum_per_mip",
                       "marchray_lo_hi_cone_o3_d3_startt_numsteps_checksum_first14_last7", "controller_rays_target_measured_nextrays", "sdfdensity_sdf16_variance16_density16", "prep_step_due_skip", "lrdecay_start_interval_base_step_factor"}
