"""Drop-in Pose Estimation Model (boundary b1 of SURVEY.md section 8):
``importlib.import_module("sam6d_amd.pem.pose_estimation_model").Net(cfg.model)``."""
