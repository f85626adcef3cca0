/* oracle/ref_time_source.c -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's step logic (Firmware/project_main/GPS/acquisition.c:93,125,175,213-223 and
 * tracking.c:94) reads the 1 ms packet counter through
 *     uint32_t signal_capture_get_packet_cnt(void);          (project_main/signal_capture.h:12)
 * whose definition lives in the MCU's SPI/DMA capture driver (project_main/signal_capture.c:35, needs the
 * STM32 StdPeriph HAL, which cannot be built on x86).  The differential harness owns the clock instead:
 * the test sets `oracle_ref_packet_cnt` before every step call.  This is the only symbol defined here; it
 * carries no arithmetic of the path under test.  The hot-path primitives are pinned against
 * _ref/libref_pm.so, which is built WITHOUT this file.
 */
#include <stdint.h>

uint32_t oracle_ref_packet_cnt = 0;

uint32_t signal_capture_get_packet_cnt(void)
{
  return oracle_ref_packet_cnt;
}
