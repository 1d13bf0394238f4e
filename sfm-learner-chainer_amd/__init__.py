"""MI355X-native photometric view-synthesis loss path of SfM-Learner."""
