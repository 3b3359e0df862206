"""Per-method trainers (optimiser groups / schedules of the reference's Examples/*.py)
on top of hawkeye_amd.train.Trainer.  `python -m hawkeye_amd.examples.BCNN --config configs/BCNN_S2_synthetic.yaml`"""
