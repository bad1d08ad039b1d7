cd /root/repo
SNERF_REPO=/root/repo SNERF_PORT=29651 timeout 300 python tools/_rccl1.py > /tmp/o.txt 2> /tmp/e.txt; echo rc=$?
tail -5 /tmp/o.txt; grep -v "^$" /tmp/e.txt | tail -30 | cut -c1-300
