# Container image (reference: Dockerfile:1-8 used pytorch:0.4.1-cuda9 + redis). ROCm 7 + PyTorch-ROCm; the shm
# transport needs no redis server. Run with --device=/dev/kfd --device=/dev/dri --ipc=host --shm-size=16g.
FROM rocm/pytorch:rocm7.0_ubuntu22.04_py3.10_pytorch_release_2.8.0
WORKDIR /opt/distributed_rl
COPY . .
ENV PYTORCH_ROCM_ARCH=gfx950 HSA_ENABLE_IPC_MODE_LEGACY=0
RUN python -c "import __graft_entry__ as g; g.build()"
CMD ["./run.sh", "4"]
