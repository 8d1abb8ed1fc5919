"""CPU restatement of the AprilRobotics detector -- TEST INFRASTRUCTURE ONLY (see apriltag_oracle.h)."""
