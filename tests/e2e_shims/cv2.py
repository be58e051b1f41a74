"""`import cv2` stand-in (train.py:15 imports it and never calls it)."""
