// pcl/registration/correspondence_rejection_median_distance.h - see correspondence_rejection.h
#include <pcl/registration/correspondence_rejection.h>
