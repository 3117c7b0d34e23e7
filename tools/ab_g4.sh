for cfg in "--scenario anisotropy --record none --steps 3" "--scenario anisotropy --record none --total-rays 1048576 --emulate-world 8 --steps 5" "--scenario anisotropy --record none --total-rays 1048576 --emulate-world 4 --steps 3" "--scenario anisotropy --method 10 --record none --rays 524288 --steps 3"; do
  for lib in raytracing_amd/librtmi.so build/librtmi_old.so raytracing_amd/librtmi.so build/librtmi_old.so; do
    echo -n "$(basename $lib) : "
    RTMI_LIB_PATH=$PWD/$lib python3 tools/bench_line.py $cfg
  done
done
