"""MI355X-native HRNet keypoint-heatmap inference path for the ESA/Kelvins SPEED pipeline."""
