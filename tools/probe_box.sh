set -x
nproc; free -g; df -h / /tmp /dev/shm $GRAFT_REPO_ROOT 2>&1
ulimit -a | head -20
which mpirun; ls /opt/rocm/include/rccl 2>&1 | head
# disk write bandwidth
( time dd if=/dev/zero of=/tmp/dd.bin bs=64M count=64 oflag=direct 2>&1 ) 2>&1 | tail -5
( time dd if=/dev/zero of=/tmp/dd2.bin bs=64M count=64 2>&1 ) 2>&1 | tail -5
rm -f /tmp/dd.bin /tmp/dd2.bin
( time dd if=/dev/zero of=/dev/shm/dd.bin bs=64M count=64 2>&1 ) 2>&1 | tail -5
rm -f /dev/shm/dd.bin
rocm-smi --showmeminfo vram 2>&1 | head -8
