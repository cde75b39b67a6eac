"""Checkpoint locations, same names/values as /root/reference/configs/__init__.py:15-16.
Neither file exists in this environment (no network); the Model then starts from random
weights and says so."""
depth_pretrain_path = './pretrained_depth_ckpt/best_depth_Ours_Bilinear_inc_3_net_G.pth'
midas_pretrain_path = './pretrained_depth_ckpt/midas_cpkt.pt'
