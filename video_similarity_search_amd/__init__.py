"""
video_similarity_search_amd — MI355X (gfx950) implementation of SLIC's contrastive-training hot
path behind the reference's Python call surface (rvl-lab-utoronto/video_similarity_search):

    models.resnet.generate_model            <- models/resnet.py:436-456
    models.triplet_net.Tripletnet           <- models/triplet_net.py:7-34
    loss.triplet_loss.OnlineTripletLoss     <- loss/triplet_loss.py:86-116 ('noise_contrastive')
    loss.NCE_loss.NCEAverage/NCESoftmaxLoss <- loss/NCE_loss.py:10-88,341-352
    clustering.cluster_masks.fit_cluster    <- clustering/cluster_masks.py:38-98 ('kmeans')
    evaluate.get_distance_matrix/get_topk_acc, retrieval.topk_retrieval
                                            <- evaluate.py:208-307, iic_retrieve_clips.py:275-314
    misc.distributed_helper                 <- misc/distributed_helper.py

All arithmetic runs in hand-written HIP kernels in csrc/ behind the C ABI of include/slic_hip.h
(libslic_hip.so, loaded with ctypes).  There is no CPU fallback: importing a compute entry point
without the built library, or calling it without a gfx950 device, raises.
"""
__version__ = "0.1.0"
