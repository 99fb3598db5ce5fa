"""CPU-side guard of the GPU parity matrix: the `-m gpu` tests named below must
exist (be collected).  A commit that deletes or renames one of them fails HERE,
in the CPU suite, instead of silently shrinking the matrix the GPU box runs
(round-1 lesson: six parity tests vanished with an unrelated cleanup)."""
import os
import subprocess
import sys

ROOT = os.path.normpath(os.path.join(os.path.dirname(__file__), ".."))

# test function -> minimum number of collected parametrisations
REQUIRED = {
    # golden vectors of the real reference
    "test_gpu_smoke.py::test_smoke_entry_point": 1,
    "test_gpu_parity.py::test_single_step_golden": 80,
    "test_gpu_parity.py::test_trajectory_golden": 56,
    "test_gpu_parity.py::test_published_known_answers": 1,
    "test_gpu_parity.py::test_pde_steps_golden": 32,
    "test_gpu_rkc.py::test_pde_steps_golden_rkc": 3,
    "test_gpu_rkc.py::test_lockstep_two_shards_with_y_dependent_spectral_radius": 1,
    "test_gpu_parity.py::test_dense_output_golden": 8,
    "test_gpu_parity.py::test_bs5_interpolants_golden": 3,
    "test_gpu_parity.py::test_ckdisc_golden": 7,
    # device RHS vs the oracle
    "test_gpu_parity.py::test_device_rhs_step_sizes": 80,
    "test_gpu_parity.py::test_atol_vector_and_rejections": 8,
    "test_gpu_parity.py::test_pde_workloads": 17,
    "test_gpu_parity.py::test_device_rhs_long_trajectory": 16,
    "test_gpu_parity.py::test_builtin_rhs_bitwise": 15,
    "test_gpu_parity.py::test_ckdisc_device_rhs_matches_oracle": 1,
    "test_gpu_parity.py::test_plugin_fallback_paths_small_and_odd_grids": 4,
    # BASELINE.json configs at full size
    "test_gpu_parity.py::test_full_size_pr8_step_matches_oracle": 1,
    "test_gpu_parity.py::test_full_size_ts5_heat_step_matches_oracle": 1,
    "test_gpu_parity.py::test_full_size_pr9_heat_step_matches_oracle": 1,
    "test_gpu_parity.py::test_full_size_linearity_and_exactness": 1,
    "test_gpu_parity.py::test_full_size_three_steps_match_oracle": 2,
    "test_gpu_parity.py::test_diffusion3d_erk_fused_sweeps": 12,
    "test_gpu_parity.py::test_full_size_free_controller_with_rejections_matches_oracle": 1,
    "test_gpu_parity.py::test_assigning_the_state_keeps_the_old_derivative": 3,
    "test_gpu_parity.py::test_full_size_fused_equals_unfused": 3,
    "test_gpu_rkc.py::test_full_size_rkc_diffusion_step_matches_oracle": 1,
    # the scipy surface and the boundary
    "test_gpu_parity.py::test_solve_ivp_device_rhs_t_eval_and_events": 3,
    "test_gpu_parity.py::test_user_plugin_compiled_with_hipcc": 1,
    "test_gpu_parity.py::test_user_plugin_with_fused_entry": 3,
    "test_gpu_parity.py::test_user_compiled_chain_plugin": 2,
    "test_gpu_parity.py::test_user_defined_tableau": 1,
    "test_gpu_parity.py::test_step_limits_and_failures": 8,
    "test_gpu_parity.py::test_classes_contract": 8,
    "test_gpu_parity.py::test_nan_propagates_to_failure": 1,
    "test_gpu_scipy_surface.py::test_events": 1,
    "test_gpu_scipy_surface.py::test_t_eval": 1,
    # round 6: whole-step attempts of the pairs with an early estimate, chains through
    # the end of an FSAL step
    "test_gpu_whole_step.py::test_whole_step_attempts_equal_the_pieces_bit_for_bit": 11,
    "test_gpu_whole_step.py::test_whole_step_controller_with_early_rejections_equals_the_pieces": 5,
    "test_gpu_whole_step.py::test_whole_step_controller_matches_the_oracle": 2,
    "test_gpu_whole_step.py::test_chain_through_the_end_of_an_fsal_step_is_bit_identical": 8,
    "test_gpu_whole_step.py::test_ts5_whole_step_is_one_launch_and_runs_ahead": 1,
    "test_gpu_whole_step.py::test_full_size_three_steps_match_oracle_early_estimate_pairs": 4,
    "test_gpu_whole_step.py::test_eight_threads_construct_solvers_with_their_own_switches": 1,
    "test_gpu_whole_step.py::test_cubic_interpolant_lives_on_the_device": 3,
    "test_gpu_whole_step.py::test_state_dependent_spectral_radius_bound_downloads_only_if_it_looks": 1,
    "test_gpu_whole_step.py::test_solve_ivp_dense_output_of_ssv2stab_matches_the_oracle_on_a_large_state": 1,
    # bit-identical restructurings, each with its on/off switch
    "test_gpu_parity.py::test_blocked_accumulation_is_bit_identical": 18,
    "test_gpu_parity.py::test_chained_stages_are_bit_identical": 45,
    "test_gpu_parity.py::test_each_epilogue_kind_is_bit_identical": 40,
    "test_gpu_parity.py::test_prelaunched_first_stage_is_used_only_when_valid": 4,
    "test_gpu_rkc.py::test_rkc_chained_stage_is_bit_identical": 1,
    "test_gpu_rkc.py::test_rkc_chain_sweeps_are_bit_identical": 35,
    "test_gpu_rkc.py::test_rkc_depth5_default_path_is_bit_identical": 1,
    "test_gpu_rkc.py::test_rkc_depth5_step_matches_oracle": 1,
    "test_gpu_rkc.py::test_rkc_chain_plan_and_whole_steps": 1,
    "test_gpu_rkc.py::test_rkc_chain_sweeps_2d_are_bit_identical": 25,
    "test_gpu_rkc.py::test_rkc_chain_2d_whole_steps": 1,
    "test_gpu_rkc.py::test_rkc_chain_2d_takes_the_end_of_the_step_along": 3,
    "test_gpu_rkc.py::test_rkc_chain_takes_the_end_of_the_step_along": 5,
    "test_gpu_parity.py::test_chained_stage_sweeps_are_bit_identical": 160,
    "test_gpu_parity.py::test_rows_only_their_sweep_reads_are_restored_on_demand": 5,
    "test_gpu_parity.py::test_lazy_rows_solve_ivp_with_dense_output": 2,
    "test_gpu_parity.py::test_end_point_derivative_as_stage_zero_of_the_next_step": 7,
    "test_gpu_parity.py::test_last_chain_forms_its_own_input": 4,
    "test_gpu_parity.py::test_first_launch_ahead_of_time": 7,
    "test_gpu_parity.py::test_chain_entry_without_declared_capabilities": 1,
    "test_gpu_rkc.py::test_rkc_fused_tail_matches_unfused": 5,
    "test_gpu_parity.py::test_host_slab_mode_is_bit_identical": 10,
    "test_gpu_parity.py::test_host_slab_mode_is_faster_for_small_host_rhs": 1,
    # lock-step
    "test_gpu_parity.py::test_rccl_single_rank_lockstep": 1,
    "test_gpu_parity.py::test_lockstep_eight_shards_on_one_gpu_equal_the_concatenated_reference": 1,
    "test_gpu_parity.py::test_lockstep_total_size_changes_the_norm": 1,
    # RKC
    "test_gpu_rkc.py::test_stages_golden": 5,
    "test_gpu_rkc.py::test_published_table": 6,
    "test_gpu_rkc.py::test_published_combustion_table": 4,
    "test_gpu_rkc.py::test_power_iteration_golden": 1,
    "test_gpu_parity.py::test_lockstep_host_reducer_estimates_the_first_step_on_the_whole_batch": 1,
    "test_gpu_rkc.py::test_lockstep_two_shards_power_iteration_and_estimated_first_step": 1,
    # complex states with a device RHS
    "test_gpu_parity.py::test_complex_device_rhs_step": 32,
    "test_gpu_parity.py::test_complex_fused_equals_unfused": 8,
    "test_gpu_parity.py::test_complex_device_rhs_trajectory_golden": 8,
    "test_gpu_parity.py::test_error_estimation_complex_device_rhs": 8,
    "test_gpu_parity.py::test_user_plugin_complex_state": 1,
    # next rows
    "test_gpu_parity.py::test_device_dense_output_large_n": 8,
    "test_gpu_parity.py::test_device_h_start": 18,
    "test_gpu_parity.py::test_diffusion3d_erk_chain_sweeps_are_bit_identical": 21,
    "test_gpu_parity.py::test_diffusion3d_erk_chain_sweeps_match_oracle": 8,
    "test_gpu_parity.py::test_free_controller_with_rejections_on_the_3d_plugin_matches_oracle": 2,
    "test_gpu_parity.py::test_lazy_state_mirror_matches_the_immediate_download": 3,
    "test_gpu_parity.py::test_lazy_state_survives_an_assignment_to_the_state": 1,
    "test_gpu_parity.py::test_solve_ivp_with_deferred_states": 4,
    "test_gpu_rkc.py::test_solve_ivp_with_deferred_states_rkc": 3,
    "test_gpu_parity.py::test_esq_options_reach_the_library_through_solve_ivp": 1,
    "test_gpu_stiffness.py::test_diagnosis_matches_reference": 1,
    # csrc/esq_stencil3d.hpp with user functors: the reference's two demo problems
    "test_gpu_stencil3d.py::test_tanh_heat_rhs_matches_the_twin": 3,
    "test_gpu_stencil3d.py::test_combustion_rhs_matches_the_twin": 3,
    "test_gpu_stencil3d.py::test_explicit_pairs_on_the_generic_sweep": 8,
    "test_gpu_stencil3d.py::test_chebyshev_stage_entry_on_the_generic_sweep": 2,
    "test_gpu_stencil3d.py::test_user_functor_with_the_fast_sweeps_and_chain_sweeps": 3,
    "test_gpu_stencil3d.py::test_published_heat_table_device_rhs": 6,
    "test_gpu_stencil3d.py::test_published_combustion_table_device_rhs": 4,
}


def test_required_gpu_tests_are_collected():
    res = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"),
         "--collect-only", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
        capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    counts = {}
    for line in res.stdout.splitlines():
        if "::" not in line:
            continue
        node = line.strip().split("[")[0]
        node = node[node.index("tests/") + len("tests/"):] if "tests/" in node else node
        counts[node] = counts.get(node, 0) + 1
    missing = {k: (counts.get(k, 0), v) for k, v in REQUIRED.items()
               if counts.get(k, 0) < v}
    assert not missing, f"GPU parity tests missing or shrunk (have, need): {missing}"
