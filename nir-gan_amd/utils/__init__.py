"""Drop-in ``utils`` package: remote_sensing_indices (HIP losses) and the YAML config reader."""
