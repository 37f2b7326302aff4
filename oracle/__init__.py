"""CPU oracle of the YOLOv4 hot path: TEST INFRASTRUCTURE ONLY (see yolov4_oracle.py)."""
