"""xformers stub: imported by hallo/models/motion_module.py:58-59, never called at inference."""
