"""Host-side mirror of the reference's ``slam`` plugin surface for the hot path
(Algorithm / Model / Optimizers / Frame) — same class and hook
names, argument meaning and error behaviour; the bodies call the HIP engine.
Only what the tracking/mapping path needs is mirrored (no CLI, datasets,
viewer, mesher)."""
