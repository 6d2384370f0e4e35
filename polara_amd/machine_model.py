"""The handful of machine constants the cost models of the package use (solver.choose_method, the two-panel exchange of
solver.ItemRows.product), in ONE table with where each number comes from.  Measured numbers cite the record under
profiles/; the multi-GPU ones are ASSUMPTIONS — no run on more than one GPU has been possible — and say so."""

MI355X = {
    # bytes of dense-row gathers per second the row-wise SpMM sustains on one GPU (A X + A^T Y of a Gramian step:
    # nnz * l * 16 bytes): 2e7 entries * 64 columns * 16 B in 1.2-1.4 ms
    'spmm_gather_Bps': (15e12, 'profiles/r04_narrow_probe_ml20m.json, r04_solver_timeline_lanczos.txt'),
    # fp64 matrix-core rate the tall-skinny kernels (gram / tsmm) reach on 26 744-row operands
    'dense_f64_flops': (20e12, 'profiles/r04_solver_timeline_lanczos.txt (tsmm 53 us for 26 744 x 448 x 64)'),
    # one nested eigen-solve of a 896 x 896 projected problem, block 64
    'nested_solve_s': (1.5e-3, 'profiles/r05_solver_timeline_lanczos.txt (a look on the main stream: 2.5-4 ms of dependent small kernels, half of it hidden)'),
    # what a block Lanczos step costs besides its sparse products and the basis traffic: ~20 small dependent launches (Gram
    # products, CholeskyQR3 with re-projection) from the library's recurrence — 24 steps: 15.0 ms of which SpMM 8.6
    'lanczos_step_fixed_s': (0.3e-3, 'profiles/r06_solver_timeline_lanczos_sync_looks.txt'),
    # wall time of a nested solve that runs on a side stream next to the products (the lag of the monitors)
    'look_wall_s': (5e-3, 'profiles/r06_krylov_block_ml20m.txt (monitor waits 4-8 ms at lag 4 x 0.55 ms steps)'),
    # bus bandwidth of a ring exchange over xGMI, per rank — ASSUMED (7 links x ~153 GB/s peak; a ring is bound by one link
    # pair): never measured, there has been no multi-GPU box
    'xgmi_bus_Bps': (100e9, 'ASSUMED: no N > 1 run over RCCL exists (SCALE_r01..r05 skipped)'),
    # latency of one small collective step — ASSUMED likewise
    'collective_step_s': (5e-6, 'ASSUMED: no N > 1 run over RCCL exists'),
}


def value(name, machine=MI355X):
    return machine[name][0]
